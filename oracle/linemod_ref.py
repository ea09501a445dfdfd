"""Oracle: numpy restatement of the LineMOD frame loader's arithmetic.  TEST INFRASTRUCTURE (tests/ only).

Follows datasets/linemod/dataset.py: ``get_bbox`` :235-277, ``mask_to_bbox`` :216-232, ``ply_vtx`` :280-291 and the
per-frame block of ``__getitem__`` :90-195 (modes 'test' / 'eval', add_noise False).  Deliberate differences, the
same two as the YCB preparation oracle (oracle/preprocess_ref.py):
  * the random pixel subset (np.random.shuffle of a 0/1 mask, :139-143) follows the documented key rule of
    include/dfusion.h instead -- a GPU cannot share numpy's Mersenne stream;
  * the model-point subset (random.sample deletion, :167-170) is passed in as the list of kept rows.
``mask_to_bbox`` calls cv2.findContours in the reference; OpenCV is absent here, so the largest bounding rectangle
is taken over the 8-connected components (scipy.ndimage.label), whose outer borders are what findContours traces
-- parity of this one helper vs OpenCV is unpinned (no fixture in the reference covers it).
"""
from __future__ import annotations

import numpy as np
import numpy.ma as ma
from scipy import ndimage

from .preprocess_ref import mix32

BORDER_LIST = [-1, 40, 80, 120, 160, 200, 240, 280, 320, 360, 400, 440, 480, 520, 560, 600, 640, 680]
CAM = dict(cx=325.26110, cy=242.04899, fx=572.41140, fy=573.57043)          # dataset.py:73-76


def mask_to_bbox(mask):
    # dataset.py:216-232: [x, y, w, h] of the contour with the largest w*h (strict '>', first wins)
    lab, n = ndimage.label(mask.astype(np.uint8), structure=np.ones((3, 3), dtype=int))
    x = y = w = h = 0
    for sl in ndimage.find_objects(lab):
        tmp_y, tmp_x = sl[0].start, sl[1].start
        tmp_h, tmp_w = sl[0].stop - sl[0].start, sl[1].stop - sl[1].start
        if tmp_w * tmp_h > w * h:
            x, y, w, h = tmp_x, tmp_y, tmp_w, tmp_h
    return [x, y, w, h]


def get_bbox(bbox):
    # dataset.py:235-277
    bbx = [bbox[1], bbox[1] + bbox[3], bbox[0], bbox[0] + bbox[2]]
    if bbx[0] < 0:
        bbx[0] = 0
    if bbx[1] >= 480:
        bbx[1] = 479
    if bbx[2] < 0:
        bbx[2] = 0
    if bbx[3] >= 640:
        bbx[3] = 639
    rmin, rmax, cmin, cmax = bbx[0], bbx[1], bbx[2], bbx[3]
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
    return rmin, rmax, cmin, cmax


def ply_vtx(path):
    # dataset.py:280-291
    with open(path) as f:
        assert f.readline().strip() == "ply"
        f.readline()
        f.readline()
        n = int(f.readline().split()[-1])
        while f.readline().strip() != "end_header":
            continue
        pts = [np.float32(f.readline().split()[:3]) for _ in range(n)]
    return np.array(pts)


def get_item(rgb, depth, label, mode, meta, model_pts_mm, keep_rows, num, seed, choose_given=None):
    """One frame (dataset.py:90-195).  rgb [480,640,3+] u8, depth [480,640] u16, label: 'eval' [480,640] u8,
    otherwise the [480,640,3] mask image.  meta: the gt.yml entry.  -> (cloud, choose, img, target, model_points)
    or None when no mask pixel lies in the crop (:135-137)."""
    mask_depth = ma.getmaskarray(ma.masked_not_equal(depth, 0))
    if mode == "eval":
        mask_label = ma.getmaskarray(ma.masked_equal(label, np.array(255)))
    else:
        mask_label = ma.getmaskarray(ma.masked_equal(label, np.array([255, 255, 255])))[:, :, 0]
    mask = mask_label * mask_depth
    img = np.transpose(np.array(rgb)[:, :, :3], (2, 0, 1))
    if mode == "eval":
        rmin, rmax, cmin, cmax = get_bbox(mask_to_bbox(mask_label))
    else:
        rmin, rmax, cmin, cmax = get_bbox(meta["obj_bb"])
    img_masked = img[:, rmin:rmax, cmin:cmax]
    target_r = np.resize(np.array(meta["cam_R_m2c"]), (3, 3))
    target_t = np.array(meta["cam_t_m2c"])
    choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero()[0]
    if len(choose) == 0:
        return None
    if choose_given is not None:           # the pixel subset as an input (the reference's own np.random.shuffle draw, tests/golden)
        choose = np.asarray(choose_given).reshape(-1).astype(np.int64)
    elif len(choose) > num:
        keys = mix32(seed, choose)
        order = np.lexsort((choose, keys))[:num]
        choose = np.sort(choose[order])
    else:
        choose = np.pad(choose, (0, num - len(choose)), "wrap")
    xmap = np.array([[j for i in range(640)] for j in range(480)])
    ymap = np.array([[i for i in range(640)] for j in range(480)])
    depth_masked = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    xmap_masked = xmap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    ymap_masked = ymap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    pt2 = depth_masked / np.float32(1.0)
    pt0 = (ymap_masked - np.float32(CAM["cx"])) * pt2 / np.float32(CAM["fx"])
    pt1 = (xmap_masked - np.float32(CAM["cy"])) * pt2 / np.float32(CAM["fy"])
    cloud = np.concatenate((pt0, pt1, pt2), axis=1)
    cloud = cloud / np.float32(1000.0)
    model_points = (model_pts_mm / 1000.0)[np.asarray(keep_rows)]
    target = np.dot(model_points, target_r.T)
    target = np.add(target, target_t / 1000.0)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    img_n = (img_masked.astype(np.float32) - mean) / std
    return (cloud.astype(np.float32), np.array([choose]).astype(np.int64), img_n, target.astype(np.float32),
            model_points.astype(np.float32), (rmin, rmax, cmin, cmax))
