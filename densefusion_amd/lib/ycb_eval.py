"""YCB-Video pose evaluation: ADD / ADD-S distances on the device, AUC on the host.

Replaces the MATLAB functions the reference drops into the YCB_Video_toolbox
(replace_ycb_toolbox/evaluate_poses_keyframe.m:160-193 ``add`` / ``adi``;
plot_accuracy_keyframe.m:29,41-53,150-170 accuracy curve, ``VOCap`` AUC up to 0.1 m, <2 cm rate).
The nearest-neighbour search of ``adi`` (every ground-truth point -> nearest estimated point) runs as a
brute-force fp64 kernel (``df_ycb_distances``), exact like MATLAB's KDTreeSearcher.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .transformations import quaternion_matrix


def pose_to_rt(pose7):
    """[q_wxyz, t] (a row of the reference's result .mat, tools/eval_ycb.py:231,239) -> 3x4 [R|t] fp64."""
    pose7 = np.asarray(pose7, dtype=np.float64)
    rt = quaternion_matrix(pose7[:4])[:3, :]
    rt[:, 3] = pose7[4:7]
    return rt


def ycb_distances(rt_est, rt_gt, pts):
    """rt_est, rt_gt [B,3,4] fp64, pts [B,M,3] fp64 (device tensors) -> (add [B], adi [B]) fp64."""
    rt_est, rt_gt, pts = (t.detach().double().contiguous() for t in (rt_est, rt_gt, pts))
    if not (rt_est.is_cuda and rt_gt.is_cuda and pts.is_cuda):
        raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
    B, M = pts.shape[0], pts.shape[1]
    add = torch.empty(B, dtype=torch.float64, device=pts.device)
    adi = torch.empty(B, dtype=torch.float64, device=pts.device)
    with _lib.device_guard(pts.device):
        st = _lib.lib().df_ycb_distances(rt_est.data_ptr(), rt_gt.data_ptr(), pts.data_ptr(), B, M, add.data_ptr(),
                                         adi.data_ptr(), _lib.current_stream())
    _lib.check(st, "ycb_distances")
    return add, adi


def voc_ap(rec, prec):
    rec, prec = np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64)
    keep = np.isfinite(rec)
    rec, prec = rec[keep], prec[keep]
    if prec.size == 0:
        return 0.0
    mrec = np.concatenate([[0.0], rec, [0.1]])
    mpre = np.maximum.accumulate(np.concatenate([[0.0], prec, [prec[-1]]]))
    i = np.flatnonzero(mrec[1:] != mrec[:-1]) + 1
    return float(np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10.0)


def auc_and_lt2cm(distances, max_distance=0.1):
    """Area under the accuracy-threshold curve up to `max_distance` (x100 = the README's AUC) and <2 cm rate."""
    D = np.array(distances, dtype=np.float64)
    D[D > max_distance] = np.inf
    d = np.sort(D)
    n = d.size
    return voc_ap(d, np.arange(1, n + 1) / n), float(np.count_nonzero(d < 0.02)) / n
