"""Frame loaders that feed the device-side input preparation."""
