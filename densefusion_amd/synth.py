"""Seeded synthetic weights and inputs for the DenseFusion hot path.

No trained checkpoint or dataset frame exists offline (reference download.sh:3-27 fetches
them), so parity tests, goldens and bench.py all run on deterministic synthetic tensors that
any box can regenerate from a seed.  Two things live here:

* the checkpoint *layout*: the (key, shape) list of ``PoseNet.state_dict()`` and
  ``PoseRefineNet.state_dict()`` exactly as the reference registers them
  (lib/network.py:27-37,39-51,70-93,136-149,170-185; lib/pspnet.py:7-18,27-34,40-62;
  lib/extractors.py:14-43,78-112) -- this is the drop-in checkpoint contract;
* numpy ``PCG64`` generators for weights and for the per-object inputs in the shapes the
  callers produce (tools/eval_ycb.py:150-190, datasets/ycb/dataset.py:227-232).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

# ---------------------------------------------------------------------------------------------
# checkpoint layout
# ---------------------------------------------------------------------------------------------
_CNN = "cnn.model.module."      # ModifiedResnet.model is wrapped in DataParallel -> ".module."


def _resnet18_spec():
    spec = [(_CNN + "feats.conv1.weight", (64, 3, 7, 7))]
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for blk in range(2):
            cin = inpl if blk == 0 else planes
            base = f"{_CNN}feats.layer{li}.{blk}."
            spec.append((base + "conv1.weight", (planes, cin, 3, 3)))
            spec.append((base + "conv2.weight", (planes, planes, 3, 3)))
            if blk == 0 and cin != planes:
                spec.append((base + "downsample.0.weight", (planes, cin, 1, 1)))
        inpl = planes
    return spec


def posenet_spec(num_obj: int):
    """(key, shape) list of PoseNet.state_dict(), in registration order."""
    spec = _resnet18_spec()
    for s in range(4):
        spec.append((f"{_CNN}psp.stages.{s}.1.weight", (512, 512, 1, 1)))
    spec += [(_CNN + "psp.bottleneck.weight", (1024, 2560, 1, 1)),
             (_CNN + "psp.bottleneck.bias", (1024,))]
    for name, cin, cout in (("up_1", 1024, 256), ("up_2", 256, 64), ("up_3", 64, 64)):
        spec += [(f"{_CNN}{name}.conv.1.weight", (cout, cin, 3, 3)),
                 (f"{_CNN}{name}.conv.1.bias", (cout,)),
                 (f"{_CNN}{name}.conv.2.weight", (1,))]          # PReLU, one shared slope
    spec += [(_CNN + "final.0.weight", (32, 64, 1, 1)), (_CNN + "final.0.bias", (32,)),
             # dead weights (PSPNet.classifier is never called) -- must still load
             (_CNN + "classifier.0.weight", (256, 256)), (_CNN + "classifier.0.bias", (256,)),
             (_CNN + "classifier.2.weight", (21, 256)), (_CNN + "classifier.2.bias", (21,))]
    for name, cin, cout in (("conv1", 3, 64), ("conv2", 64, 128), ("e_conv1", 32, 64),
                            ("e_conv2", 64, 128), ("conv5", 256, 512), ("conv6", 512, 1024)):
        spec += [(f"feat.{name}.weight", (cout, cin, 1)), (f"feat.{name}.bias", (cout,))]
    for layer, cin, cout in ((1, 1408, 640), (2, 640, 256), (3, 256, 128)):
        for h in "rtc":
            spec += [(f"conv{layer}_{h}.weight", (cout, cin, 1)), (f"conv{layer}_{h}.bias", (cout,))]
    for h, per in (("r", 4), ("t", 3), ("c", 1)):
        spec += [(f"conv4_{h}.weight", (num_obj * per, 128, 1)), (f"conv4_{h}.bias", (num_obj * per,))]
    return spec


def refiner_spec(num_obj: int):
    """(key, shape) list of PoseRefineNet.state_dict(), in registration order."""
    spec = []
    for name, cin, cout in (("conv1", 3, 64), ("conv2", 64, 128), ("e_conv1", 32, 64),
                            ("e_conv2", 64, 128), ("conv5", 384, 512), ("conv6", 512, 1024)):
        spec += [(f"feat.{name}.weight", (cout, cin, 1)), (f"feat.{name}.bias", (cout,))]
    for layer, cin, cout in ((1, 1024, 512), (2, 512, 128)):
        for h in "rt":
            spec += [(f"conv{layer}_{h}.weight", (cout, cin)), (f"conv{layer}_{h}.bias", (cout,))]
    for h, per in (("r", 4), ("t", 3)):
        spec += [(f"conv3_{h}.weight", (num_obj * per, 128)), (f"conv3_{h}.bias", (num_obj * per,))]
    return spec


def num_params(spec) -> int:
    return int(sum(int(np.prod(s)) for _, s in spec))


# ---------------------------------------------------------------------------------------------
# synthetic weights
# ---------------------------------------------------------------------------------------------
# output-layer gains: keep log-softmax logits O(1), translation offsets at the centimetre scale and
# the confidence sigmoid unsaturated, so that arg-max selection and ADD values are well conditioned
_KEY_GAIN = (("final.0.weight", 0.04), ("conv4_t.weight", 0.004), ("conv4_c.weight", 0.6),
             ("conv3_t.weight", 0.01), ("conv1_r.weight", 0.5), ("conv1_t.weight", 0.5),
             ("conv1_c.weight", 0.5))


def make_state_dict(spec, seed: int, gain: float = 1.0):
    """Fill every tensor of ``spec`` in key order from one PCG64 stream (numpy float32).

    Weights ~ N(0, gain*sqrt(2/fan_in)) (keeps activation scale roughly constant through the
    ReLU stack, which has no normalisation layers); biases ~ U(-0.05, 0.05); PReLU slope 0.25.
    The stem is scaled down so that the 0..255-scale normalised image the reference feeds
    (tools/eval_ycb.py:33,181) lands at O(1) after conv1.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for key, shape in spec:
        if key.endswith(".conv.2.weight"):
            sd[key] = np.full(shape, 0.25, dtype=np.float32)
        elif key.endswith(".bias"):
            sd[key] = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            std = gain * math.sqrt(2.0 / fan_in)
            if key.endswith("feats.conv1.weight"):
                std *= 1.0 / 400.0
            if ".downsample." in key or ".conv2.weight" in key and ".layer" in key:
                std *= math.sqrt(0.5)      # two branches are summed in every BasicBlock
            for pat, g in _KEY_GAIN:
                if key.endswith(pat):
                    std *= g
            wgt = rng.standard_normal(size=shape) * std
            if len(shape) == 3 and shape[1] == 1408:
                wgt[:, 384:] *= 0.1     # damp the broadcast global feature so per-point outputs differ
            sd[key] = wgt.astype(np.float32)
    return sd


# ---------------------------------------------------------------------------------------------
# synthetic per-object inputs
# ---------------------------------------------------------------------------------------------
YCB_CAM = dict(cx=312.9869, cy=241.3109, fx=1066.778, fy=1067.487)       # eval_ycb.py:37-41
LINEMOD_CAM = dict(cx=325.26110, cy=242.04899, fx=572.41140, fy=573.57043)  # linemod/dataset.py:73-76
_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def random_unit_quaternion(rng) -> np.ndarray:
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    return q


def quat_to_rot(q) -> np.ndarray:
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_object(seed: int, H: int, W: int, num_points: int, num_obj: int,
                num_points_mesh: int = 500, cam=YCB_CAM):
    """One synthetic (frame, object) sample in the reference's dataset tuple layout.

    Returns dict with img[3,H,W] f32, cloud[N,3] f32, choose[1,N] i64, obj[1] i64,
    model_points[M,3] f32, target[M,3] f32 (datasets/ycb/dataset.py:227-232).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    # the reference normalises un-scaled 0..255 values with ImageNet mean/std
    rgb = rng.integers(0, 256, size=(3, H, W)).astype(np.float32)
    img = (rgb - _MEAN[:, None, None]) / _STD[:, None, None]
    # choose: N pixel indices of a ~40 %-fill mask (random subset, or wrap-padded when too few)
    mask = np.flatnonzero(rng.random(H * W) < 0.4)
    if len(mask) > num_points:
        sel = np.zeros(len(mask), dtype=bool)
        sel[:num_points] = True
        rng.shuffle(sel)
        choose = mask[sel]
    else:
        choose = np.pad(mask, (0, num_points - len(mask)), "wrap")
    choose = choose.astype(np.int64)
    # back-projected cloud
    row0 = int(rng.integers(0, 480 - H + 1)) if H <= 480 else 0
    col0 = int(rng.integers(0, 640 - W + 1)) if W <= 640 else 0
    v = (choose // W + row0).astype(np.float32)
    u = (choose % W + col0).astype(np.float32)
    z = rng.uniform(0.5, 1.2, size=num_points).astype(np.float32)
    x = (u - np.float32(cam["cx"])) * z / np.float32(cam["fx"])
    y = (v - np.float32(cam["cy"])) * z / np.float32(cam["fy"])
    cloud = np.stack([x, y, z], axis=1).astype(np.float32)
    obj = np.array([int(rng.integers(0, num_obj))], dtype=np.int64)
    ext = rng.uniform(0.1, 0.25, size=3)
    model_points = ((rng.random((num_points_mesh, 3)) - 0.5) * ext).astype(np.float32)
    Rgt = quat_to_rot(random_unit_quaternion(rng))
    tgt = cloud.mean(axis=0).astype(np.float64)
    target = (model_points.astype(np.float64) @ Rgt.T + tgt).astype(np.float32)
    return dict(img=img, cloud=cloud, choose=choose[None, :], obj=obj,
                model_points=model_points, target=target)


def make_batch(seed: int, B: int, H: int, W: int, num_points: int, num_obj: int,
               num_points_mesh: int = 500, cam=YCB_CAM):
    """B same-size objects stacked on a leading axis (the build's batched extension)."""
    objs = [make_object(seed * 1000 + i, H, W, num_points, num_obj, num_points_mesh, cam)
            for i in range(B)]
    return {k: np.stack([o[k] for o in objs]) for k in objs[0]}


# ---- SegNet (vanilla_segmentation/segnet.py): state-dict layout and seeded synthetic weights -------------------------
_SEG_LAYERS = [("11", 3, 64), ("12", 64, 64), ("21", 64, 128), ("22", 128, 128), ("31", 128, 256), ("32", 256, 256), ("33", 256, 256),
               ("41", 256, 512), ("42", 512, 512), ("43", 512, 512), ("51", 512, 512), ("52", 512, 512), ("53", 512, 512),
               ("53d", 512, 512), ("52d", 512, 512), ("51d", 512, 512), ("43d", 512, 512), ("42d", 512, 512), ("41d", 512, 256),
               ("33d", 256, 256), ("32d", 256, 256), ("31d", 256, 128), ("22d", 128, 128), ("21d", 128, 64), ("12d", 64, 64), ("11d", 64, None)]


def segnet_spec(label_nbr=22, input_nbr=3):
    """(key, shape) in the reference's state_dict order (conv then its bn, constructor order of segnet.py:12-70)."""
    spec = []
    for name, cin, cout in _SEG_LAYERS:
        cin = input_nbr if name == "11" else cin
        cout = label_nbr if cout is None else cout
        spec += [(f"conv{name}.weight", (cout, cin, 3, 3)), (f"conv{name}.bias", (cout,))]
        if name != "11d":
            spec += [(f"bn{name}.weight", (cout,)), (f"bn{name}.bias", (cout,)), (f"bn{name}.running_mean", (cout,)),
                     (f"bn{name}.running_var", (cout,)), (f"bn{name}.num_batches_tracked", ())]
    return spec


def make_segnet_state_dict(seed, label_nbr=22):
    """Seeded synthetic SegNet weights: He-normal convs, BatchNorm scale / variance around 1, small shifts / means."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape in segnet_spec(label_nbr):
        if key.endswith("num_batches_tracked"):
            sd[key] = np.array(100, dtype=np.int64)
        elif key.startswith("conv") and key.endswith("weight"):
            fan_in = shape[1] * 9
            sd[key] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif key.endswith("running_var") or (key.startswith("bn") and key.endswith("weight")):
            sd[key] = rng.uniform(0.6, 1.4, shape).astype(np.float32)
        else:
            sd[key] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
    return sd
