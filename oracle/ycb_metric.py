"""Oracle: numpy fp64 restatement of the YCB-Video toolbox evaluation.  TEST INFRASTRUCTURE.

Follows replace_ycb_toolbox/evaluate_poses_keyframe.m:150-193 (``transform_pts_Rt``, ``add``, ``adi`` --
``adi`` searches, for every GROUND-TRUTH point, the nearest ESTIMATED point with a KD-tree) and
replace_ycb_toolbox/plot_accuracy_keyframe.m:29,41-53 (threshold 0.1 m, accuracy curve, <2 cm rate) and
:150-170 (``VOCap``).  MATLAB cannot run here and the toolbox is a network fetch: parity unpinned.
"""
from __future__ import annotations

import numpy as np


def transform_pts_Rt(pts, RT):
    """pts [3,n], RT [3,4] -> [3,n]   (evaluate_poses_keyframe.m:150-158)."""
    pts = np.asarray(pts, dtype=np.float64)
    return np.asarray(RT, dtype=np.float64) @ np.vstack([pts, np.ones((1, pts.shape[1]))])


def add(RT_est, RT_gt, pts):
    d = transform_pts_Rt(pts, RT_est) - transform_pts_Rt(pts, RT_gt)
    return float(np.mean(np.sqrt(np.sum(d ** 2, axis=0))))


def adi(RT_est, RT_gt, pts):
    e, g = transform_pts_Rt(pts, RT_est), transform_pts_Rt(pts, RT_gt)
    best = np.full(g.shape[1], np.inf)
    for s in range(0, e.shape[1], 512):                       # exact brute force == exact KD-tree search
        d = ((e[:, s:s + 512, None] - g[:, None, :]) ** 2).sum(0)
        best = np.minimum(best, d.min(0))
    return float(np.mean(np.sqrt(best)))


def voc_ap(rec, prec):
    """plot_accuracy_keyframe.m:150-170."""
    rec, prec = np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64)
    keep = np.isfinite(rec)
    rec, prec = rec[keep], prec[keep]
    if prec.size == 0:
        return 0.0
    mrec = np.concatenate([[0.0], rec, [0.1]])
    mpre = np.concatenate([[0.0], prec, [prec[-1]]])
    for i in range(1, mpre.size):
        mpre[i] = max(mpre[i], mpre[i - 1])
    i = np.flatnonzero(mrec[1:] != mrec[:-1]) + 1
    return float(np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10.0)


def auc_and_lt2cm(distances, max_distance=0.1):
    """plot_accuracy_keyframe.m:41-53 -> (AUC in [0,1], fraction < 2 cm)."""
    D = np.array(distances, dtype=np.float64)
    D[D > max_distance] = np.inf
    d = np.sort(D)
    n = d.size
    accuracy = np.cumsum(np.ones(n)) / n
    return voc_ap(d, accuracy), float(np.count_nonzero(d < 0.02)) / n
