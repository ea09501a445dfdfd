"""Oracle: host-side pose algebra of the eval loops (numpy float64).  TEST INFRASTRUCTURE.

Follows lib/transformations.py:1254-1278 (quaternion_matrix), :1320-1341,1361-1363
(quaternion_from_matrix with isprecise=True), tools/eval_ycb.py:192-229 /
tools/eval_linemod.py:81-114 (per-pixel pose selection + iterative refinement) and
tools/eval_linemod.py:118-130 (ADD / ADD-S metric).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import dfnet

_EPS = np.finfo(float).eps * 4.0       # lib/transformations.py:1893


def quaternion_matrix(quaternion) -> np.ndarray:
    # lib/transformations.py:1266-1278
    q = np.array(quaternion, dtype=np.float64, copy=True)
    n = float(np.dot(q, q))
    if n < _EPS:
        return np.identity(4)
    q *= math.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([
        [1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
        [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
        [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
        [0.0, 0.0, 0.0, 1.0]])


def quaternion_from_matrix_precise(matrix) -> np.ndarray:
    # lib/transformations.py:1320-1341 (isprecise branch) + sign normalisation :1361-1363
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    q = np.empty((4,))
    t = np.trace(M)
    if t > M[3, 3]:
        q[0] = t
        q[3] = M[1, 0] - M[0, 1]
        q[2] = M[0, 2] - M[2, 0]
        q[1] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
        q = q[[3, 0, 1, 2]]
    q *= 0.5 / math.sqrt(t * M[3, 3])
    if q[0] < 0.0:
        np.negative(q, q)
    return q


def select_pose(pred_r, pred_t, pred_c, cloud):
    """tools/eval_ycb.py:193-203 -- normalise quats, argmax confidence, (q, points+t) at it.

    Inputs are torch tensors [1,N,4],[1,N,3],[1,N,1],[1,N,3]; returns float32 numpy (4,),(3,), int.
    """
    n = pred_r.shape[1]
    pred_r = pred_r / torch.norm(pred_r, dim=2).view(1, n, 1)
    which = int(torch.max(pred_c.view(1, n), 1)[1][0])
    my_r = pred_r[0][which].view(-1).numpy()
    my_t = (cloud.view(n, 1, 3) + pred_t.view(n, 1, 3))[which].view(-1).numpy()
    return my_r, my_t, which


def refine_step(sd_ref, cloud, emb, obj, my_r, my_t):
    """One pass of the refine loop body, tools/eval_ycb.py:206-229."""
    n = cloud.shape[1]
    dt = cloud.dtype
    T = torch.from_numpy(np.asarray(my_t).astype(np.float32)).to(dt).view(1, 3).repeat(n, 1).view(1, n, 3)
    my_mat = quaternion_matrix(my_r)
    R = torch.from_numpy(my_mat[:3, :3].astype(np.float32)).to(dt).view(1, 3, 3)
    my_mat[0:3, 3] = my_t
    new_cloud = torch.bmm(cloud - T, R).contiguous()
    pred_r, pred_t = dfnet.refiner_forward(sd_ref, new_cloud, emb, obj)
    pred_r = pred_r.view(1, 1, -1)
    pred_r = pred_r / torch.norm(pred_r, dim=2).view(1, 1, 1)
    my_r_2 = pred_r.view(-1).numpy()
    my_t_2 = pred_t.view(-1).numpy()
    my_mat_2 = quaternion_matrix(my_r_2)
    my_mat_2[0:3, 3] = my_t_2
    final = np.dot(my_mat, my_mat_2)
    rot = final.copy()
    rot[0:3, 3] = 0
    return quaternion_from_matrix_precise(rot), np.array([final[0][3], final[1][3], final[2][3]])


def estimate_pose(sd_pose, sd_ref, img, cloud, choose, obj, iteration):
    """PoseNet -> select -> ``iteration`` refine steps.  Returns (pose_wo_refine[7], pose[7])."""
    pred_r, pred_t, pred_c, emb = dfnet.posenet_forward(sd_pose, img, cloud, choose, obj)
    my_r, my_t, _ = select_pose(pred_r, pred_t, pred_c, cloud)
    wo = np.append(my_r, my_t).astype(np.float64)
    for _ in range(iteration):
        my_r, my_t = refine_step(sd_ref, cloud, emb, obj, my_r, my_t)
    return wo, np.append(my_r, my_t).astype(np.float64)


def transform_model(pose7, model_points) -> np.ndarray:
    # tools/eval_linemod.py:118-121
    R = quaternion_matrix(pose7[:4])[:3, :3]
    return np.dot(np.asarray(model_points, dtype=np.float64), R.T) + np.asarray(pose7[4:7], dtype=np.float64)


def add_metric(pred, target) -> float:
    # tools/eval_linemod.py:130 -- mean point-to-point distance
    return float(np.mean(np.linalg.norm(np.asarray(pred) - np.asarray(target), axis=1)))


def adds_metric(pred, target) -> float:
    """ADD-S in the LineMOD direction (each pred point -> nearest target point),
    tools/eval_linemod.py:123-128; float32 like the reference's .cuda() tensors."""
    from .knn import knn_ref
    p = np.ascontiguousarray(np.asarray(pred, dtype=np.float32).T)      # [3,Q]
    t = np.ascontiguousarray(np.asarray(target, dtype=np.float32).T)    # [3,R]
    inds = knn_ref(t[None], p[None], 1)[0, 0] - 1
    sel = t[:, inds]
    return float(np.mean(np.linalg.norm((p.T - sel.T).astype(np.float32), axis=1), dtype=np.float32))
