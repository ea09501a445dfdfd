"""Oracle: numpy restatement of the per-object input preparation.  TEST INFRASTRUCTURE.

Follows tools/eval_ycb.py:54-90 (``get_bbox``) and :150-181 (mask, choose, cloud, normalised crop).  The one
deliberate difference is the random subset: the reference shuffles a 0/1 mask with np.random
(eval_ycb.py:157-161); build and oracle share the documented key rule instead (include/dfusion.h:
keep the num_points mask pixels with the smallest mix32(seed, flat index), in index order).
"""
from __future__ import annotations

import numpy as np
import numpy.ma as ma

BORDER_LIST = [-1, 40, 80, 120, 160, 200, 240, 280, 320, 360, 400, 440, 480, 520, 560, 600, 640, 680]


def get_bbox(roi, img_width=480, img_length=640):
    # tools/eval_ycb.py:54-90
    rmin = int(roi[3]) + 1
    rmax = int(roi[5]) - 1
    cmin = int(roi[2]) + 1
    cmax = int(roi[4]) - 1
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
    if rmax > img_width:
        delt = rmax - img_width
        rmax = img_width
        rmin -= delt
    if cmax > img_length:
        delt = cmax - img_length
        cmax = img_length
        cmin -= delt
    return rmin, rmax, cmin, cmax


def mix32(seed, i):
    i = np.asarray(i, dtype=np.uint64)
    x = (np.uint64(seed & 0xFFFFFFFF) ^ ((i * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF))) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def prepare_object(rgb, depth, label, itemid, bbox, num_points, seed, cam):
    """rgb [IH,IW,3] u8, depth [IH,IW] u16, label [IH,IW] -> img [3,H,W] f32, cloud [N,3] f32, choose [1,N] i64, count."""
    rmin, rmax, cmin, cmax = bbox
    IH, IW = depth.shape
    xmap = np.array([[j for i in range(IW)] for j in range(IH)])
    ymap = np.array([[i for i in range(IW)] for j in range(IH)])
    mask_depth = ma.getmaskarray(ma.masked_not_equal(depth, 0))
    mask_label = ma.getmaskarray(ma.masked_equal(label, itemid))
    mask = mask_label * mask_depth
    choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero()[0]
    count = len(choose)
    if count == 0:
        return None, None, None, 0
    if count > num_points:
        keys = mix32(seed, choose)
        order = np.lexsort((choose, keys))[:num_points]        # smallest keys, ties -> lower index
        choose = np.sort(choose[order])
    else:
        choose = np.pad(choose, (0, num_points - count), "wrap")
    depth_masked = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    xmap_masked = xmap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    ymap_masked = ymap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    pt2 = depth_masked / np.float32(cam["scale"])
    pt0 = (ymap_masked - np.float32(cam["cx"])) * pt2 / np.float32(cam["fx"])
    pt1 = (xmap_masked - np.float32(cam["cy"])) * pt2 / np.float32(cam["fy"])
    cloud = np.concatenate((pt0, pt1, pt2), axis=1).astype(np.float32)
    img = np.transpose(rgb[:, :, :3], (2, 0, 1))[:, rmin:rmax, cmin:cmax].astype(np.float32)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    img = (img - mean) / std
    return img, cloud, np.array([choose]).astype(np.int64), count
