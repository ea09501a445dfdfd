"""Oracle: numpy restatement of the YCB-Video frame loader's arithmetic (real frames, add_noise False).  TEST INFRASTRUCTURE.

Follows datasets/ycb/dataset.py: ``get_bbox`` :251-289 and ``__getitem__`` :90-106,139-150,168-217.  As in the other
preparation oracles the random pixel subset follows the key rule of include/dfusion.h instead of np.random.shuffle
(:174-178); the chosen object index and the kept model rows are passed in.
"""
from __future__ import annotations

import numpy as np
import numpy.ma as ma

from .preprocess_ref import mix32

BORDER_LIST = [-1, 40, 80, 120, 160, 200, 240, 280, 320, 360, 400, 440, 480, 520, 560, 600, 640, 680]


def get_bbox(label):
    # datasets/ycb/dataset.py:251-289
    rows = np.any(label, axis=1)
    cols = np.any(label, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    rmax += 1
    cmax += 1
    r_b = rmax - rmin
    for tt in range(len(BORDER_LIST) - 1):
        if r_b > BORDER_LIST[tt] and r_b < BORDER_LIST[tt + 1]:
            r_b = BORDER_LIST[tt + 1]
            break
    c_b = cmax - cmin
    for tt in range(len(BORDER_LIST) - 1):
        if c_b > BORDER_LIST[tt] and c_b < BORDER_LIST[tt + 1]:
            c_b = BORDER_LIST[tt + 1]
            break
    center = [int((rmin + rmax) / 2), int((cmin + cmax) / 2)]
    rmin = center[0] - int(r_b / 2)
    rmax = center[0] + int(r_b / 2)
    cmin = center[1] - int(c_b / 2)
    cmax = center[1] + int(c_b / 2)
    if rmin < 0:
        delt = -rmin
        rmin = 0
        rmax += delt
    if cmin < 0:
        delt = -cmin
        cmin = 0
        cmax += delt
    if rmax > 480:
        delt = rmax - 480
        rmax = 480
        rmin -= delt
    if cmax > 640:
        delt = cmax - 640
        cmax = 640
        cmin -= delt
    return int(rmin), int(rmax), int(cmin), int(cmax)


def get_item(rgb, depth, label, meta, seq_no, idx, cld, keep_rows, num_pt, seed, choose_given=None):
    """One real frame with object slot `idx` (dataset.py:90-217) -> cloud, choose, img, target, model_points, box."""
    cam = (323.7872, 279.6921, 1077.836, 1078.189) if seq_no >= 60 else (312.9869, 241.3109, 1066.778, 1067.487)
    cam_cx, cam_cy, cam_fx, cam_fy = (np.float32(v) for v in cam)
    obj = meta["cls_indexes"].flatten().astype(np.int32)
    mask_depth = ma.getmaskarray(ma.masked_not_equal(depth, 0))
    mask_label = ma.getmaskarray(ma.masked_equal(label, obj[idx]))
    mask = mask_label * mask_depth
    rmin, rmax, cmin, cmax = get_bbox(mask_label)
    img = np.transpose(np.array(rgb)[:, :, :3], (2, 0, 1))[:, rmin:rmax, cmin:cmax]
    target_r = meta["poses"][:, :, idx][:, 0:3]
    target_t = np.array([meta["poses"][:, :, idx][:, 3:4].flatten()])
    choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero()[0]
    if choose_given is not None:           # the pixel subset as an input (the reference's own np.random.shuffle draw, tests/golden)
        choose = np.asarray(choose_given).reshape(-1).astype(np.int64)
    elif len(choose) > num_pt:
        keys = mix32(seed, choose)
        order = np.lexsort((choose, keys))[:num_pt]
        choose = np.sort(choose[order])
    else:
        choose = np.pad(choose, (0, num_pt - len(choose)), "wrap")
    xmap = np.array([[j for i in range(640)] for j in range(480)])
    ymap = np.array([[i for i in range(640)] for j in range(480)])
    depth_masked = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    xmap_masked = xmap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    ymap_masked = ymap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    cam_scale = np.float32(meta["factor_depth"][0][0])
    pt2 = depth_masked / cam_scale
    pt0 = (ymap_masked - cam_cx) * pt2 / cam_fx
    pt1 = (xmap_masked - cam_cy) * pt2 / cam_fy
    cloud = np.concatenate((pt0, pt1, pt2), axis=1)
    model_points = cld[np.asarray(keep_rows)]
    target = np.add(np.dot(model_points, target_r.T), target_t)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    img_n = (img.astype(np.float32) - mean) / std
    return (cloud.astype(np.float32), np.array([choose]).astype(np.int64), img_n, target.astype(np.float32),
            model_points.astype(np.float32), (rmin, rmax, cmin, cmax))
