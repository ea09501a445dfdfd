"""Device-side input preparation for the eval loop.

``get_bbox`` is the host integer arithmetic of tools/eval_ycb.py:54-90 (snap the detector ROI to the
border list, keep it inside the 480x640 frame); ``preprocess_objects`` runs everything the reference
then does in numpy per object (tools/eval_ycb.py:150-181) as one HIP launch over B objects of one
crop size: mask, ``choose`` sampling, depth back-projection, normalised colour crop.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib

BORDER_LIST = [-1, 40, 80, 120, 160, 200, 240, 280, 320, 360, 400, 440, 480, 520, 560, 600, 640, 680]
IMG_WIDTH, IMG_LENGTH = 480, 640            # eval_ycb.py:43-44 (rows, columns)
YCB_CAM = dict(cx=312.9869, cy=241.3109, fx=1066.778, fy=1067.487, scale=10000.0)     # eval_ycb.py:37-41
# datasets/linemod/dataset.py:73-76,152-157: back-projection in millimetres (cam_scale 1), finished cloud / 1000
IMG_MEAN, IMG_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # the kernel's normalisation (applied to 0-255-scale pixels, eval_ycb.py:33)
LINEMOD_CAM = dict(cx=325.26110, cy=242.04899, fx=572.41140, fy=573.57043, scale=1.0, cloud_div=1000.0)


def _snap(extent):
    for lo, hi in zip(BORDER_LIST[:-1], BORDER_LIST[1:]):
        if lo < extent < hi:
            return hi
    return extent


def get_bbox(roi, img_width=IMG_WIDTH, img_length=IMG_LENGTH):
    """PoseCNN roi row [batch, cls, x1, y1, x2, y2, ...] -> (rmin, rmax, cmin, cmax)  (eval_ycb.py:54-90)."""
    rmin, rmax = int(roi[3]) + 1, int(roi[5]) - 1
    cmin, cmax = int(roi[2]) + 1, int(roi[4]) - 1
    r_b, c_b = _snap(rmax - rmin), _snap(cmax - cmin)
    cr, cc = int((rmin + rmax) / 2), int((cmin + cmax) / 2)
    rmin, rmax = cr - int(r_b / 2), cr + int(r_b / 2)
    cmin, cmax = cc - int(c_b / 2), cc + int(c_b / 2)
    if rmin < 0:
        rmax, rmin = rmax - rmin, 0
    if cmin < 0:
        cmax, cmin = cmax - cmin, 0
    if rmax > img_width:
        rmin, rmax = rmin - (rmax - img_width), img_width
    if cmax > img_length:
        cmin, cmax = cmin - (cmax - img_length), img_length
    return rmin, rmax, cmin, cmax


def preprocess_objects(rgb, depth, label, objects, num_points, cam=YCB_CAM, choose_in=None):
    """rgb [F,IH,IW,3] uint8, depth [F,IH,IW] uint16 (as int16 bits is fine), label [F,IH,IW] int32 -- device tensors.
    objects: list of (frame, itemid, (rmin, rmax, cmin, cmax), seed), all boxes of one size.
    choose_in (optional, [B,N] / [B,1,N] int64): the chosen pixel indices as an INPUT (sampling skipped) -- the reference's own
    np.random.shuffle subset in the golden tests.
    Returns img [B,3,H,W], cloud [B,N,3], choose [B,1,N] int64, count [B] int32 (0 = lost detection)."""
    if not (rgb.is_cuda and depth.is_cuda and label.is_cuda):
        raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
    F, IH, IW, _ = rgb.shape
    B = len(objects)
    H = objects[0][2][1] - objects[0][2][0]
    W = objects[0][2][3] - objects[0][2][2]
    desc = np.zeros((B, 8), dtype=np.int32)
    for i, (frame, itemid, (rmin, rmax, cmin, cmax), seed) in enumerate(objects):
        if rmax - rmin != H or cmax - cmin != W:
            raise RuntimeError("preprocess_objects: all boxes of one call must have the same size")
        if not (0 <= frame < F and 0 <= rmin and rmax <= IH and 0 <= cmin and cmax <= IW):
            raise RuntimeError("preprocess_objects: box outside the frame")
        desc[i, :6] = (frame, itemid, rmin, rmax, cmin, cmax)
        desc[i, 6] = np.array([seed & 0xFFFFFFFF], dtype=np.uint32).view(np.int32)[0]      # uint32 seed bits
        desc[i, 7] = 1 if choose_in is not None else 0
    dev = rgb.device
    d_desc = torch.from_numpy(desc).pin_memory().to(dev, non_blocking=True)      # pinned: the upload does not wait for the stream
    rgb, label = rgb.contiguous(), label.to(torch.int32).contiguous()
    depth = depth.contiguous()
    if depth.dtype not in (torch.int16, torch.uint16):
        raise RuntimeError("preprocess_objects: depth must be 16-bit")
    scratch = torch.empty(B * H * W, dtype=torch.int32, device=dev)
    img = torch.empty(B, 3, H, W, device=dev)
    cloud = torch.empty(B, num_points, 3, device=dev)
    if choose_in is not None:
        choose = choose_in.to(device=dev, dtype=torch.int64).reshape(B, 1, num_points).contiguous().clone()
    else:
        choose = torch.empty(B, 1, num_points, dtype=torch.int64, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    with _lib.device_guard(dev):
        st = _lib.lib().df_preprocess_objects(rgb.data_ptr(), depth.data_ptr(), label.data_ptr(), F, IH, IW, d_desc.data_ptr(), B,
                                              H, W, num_points, cam["cx"], cam["cy"], cam["fx"], cam["fy"], cam["scale"], cam.get("cloud_div", 1.0),
                                              scratch.data_ptr(), img.data_ptr(), cloud.data_ptr(), choose.data_ptr(),
                                              count.data_ptr(), _lib.current_stream())
    _lib.check(st, "preprocess_objects")
    return img, cloud, choose, count
