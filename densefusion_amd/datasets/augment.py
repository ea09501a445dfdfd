"""Training-time augmentation of the two disk datasets (datasets/ycb/dataset.py:84,117-136,149-167,196-221;
datasets/linemod/dataset.py:83,114-115,132,159-160,178-180): colour jitter, pose-translation noise and -- YCB only -- synthetic
frames pasted over real backgrounds with occluders from other synthetic frames.

``ColorJitter`` restates ``torchvision.transforms.ColorJitter`` of torchvision 0.2.2.post3 (the reference's pin, Dockerfile:27; the
package is not part of this build): four factors drawn with Python's ``random.uniform`` in the order brightness, contrast,
saturation, hue, the four operations applied in an order given by ``random.shuffle``; brightness / contrast / saturation are PIL's
``ImageEnhance.Brightness / Contrast / Color``, hue adds ``uint8(hue_factor * 255)`` to the H plane of the HSV image with 8-bit
wrap-around.  Same draws from the same ``random`` state, same PIL calls; parity with torchvision itself is unpinned (it cannot be
imported here).  Everything in this file runs on the host, in the loader's worker processes.
"""
from __future__ import annotations

import random

import numpy as np
from PIL import Image, ImageEnhance


def adjust_hue(img, hue_factor):
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError("hue_factor is not in [-0.5, 0.5].")
    mode = img.mode
    if mode in ("L", "1", "I", "F"):
        return img
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = (np_h.astype(np.int32) + (int(hue_factor * 255) & 0xFF)).astype(np.uint8)      # uint8 addition: wraps across the boundary
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert(mode)


class ColorJitter:
    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        self.brightness = (max(0.0, 1.0 - brightness), 1.0 + brightness) if brightness else None
        self.contrast = (max(0.0, 1.0 - contrast), 1.0 + contrast) if contrast else None
        self.saturation = (max(0.0, 1.0 - saturation), 1.0 + saturation) if saturation else None
        self.hue = (-hue, hue) if hue else None

    def get_params(self):
        ops = []
        if self.brightness is not None:
            f = random.uniform(*self.brightness)
            ops.append(lambda im, f=f: ImageEnhance.Brightness(im).enhance(f))
        if self.contrast is not None:
            f = random.uniform(*self.contrast)
            ops.append(lambda im, f=f: ImageEnhance.Contrast(im).enhance(f))
        if self.saturation is not None:
            f = random.uniform(*self.saturation)
            ops.append(lambda im, f=f: ImageEnhance.Color(im).enhance(f))
        if self.hue is not None:
            f = random.uniform(*self.hue)
            ops.append(lambda im, f=f: adjust_hue(im, f))
        random.shuffle(ops)
        return ops

    def __call__(self, img):
        for op in self.get_params():
            img = op(img)
        return img


def occluder_mask(f_label, front_num=2):
    """datasets/ycb/dataset.py:122-132: `front_num` objects drawn from a synthetic frame's label image; returns the boolean mask that is
    False on those objects' pixels (None when the frame shows fewer objects)."""
    front_label = np.unique(f_label).tolist()[1:]
    if len(front_label) < front_num:
        return None
    keep = np.ones(f_label.shape, dtype=bool)
    for f_i in random.sample(front_label, front_num):
        keep &= f_label != f_i
    return keep
