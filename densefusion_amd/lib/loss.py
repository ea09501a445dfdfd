"""``Loss`` -- host-side mirror of lib/loss.py:73-82 over the fused HIP kernels (forward only).

``Loss(num_points_mesh, sym_list).forward(pred_r, pred_t, pred_c, target, model_points, idx, points, w,
refine)`` -> ``(loss, dis, new_points [1,N,3], new_target [1,M,3])`` with the reference's shapes.  The
symmetric branch implements the semantics the reference intends (lib/knn 1-NN of every transformed model
point among the target points, lib/loss.py:9,41-47) -- in this fork that branch raises because of a
mis-wired import (SURVEY header note 3).  No autograd graph is built (backward kernels: SURVEY 8f4).
"""
from __future__ import annotations

import torch

from .. import _lib


def _f32(t):
    if not t.is_cuda:
        raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
    return t.detach().float().contiguous()


class Loss:
    def __init__(self, num_points_mesh, sym_list):
        self.num_pt_mesh = int(num_points_mesh)
        self.sym_list = list(sym_list)

    def forward(self, pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine):
        pred_r, pred_t, pred_c = _f32(pred_r), _f32(pred_t), _f32(pred_c)
        target, model_points, points = _f32(target), _f32(model_points), _f32(points)
        bs, N = pred_c.shape[0], pred_c.shape[1]
        M = self.num_pt_mesh
        if bs != 1 or target.numel() != M * 3 or model_points.numel() != M * 3 or points.numel() != N * 3:
            raise RuntimeError("Loss.forward: expected bs = 1, target/model_points [1,M,3], points [1,N,3]")
        dev = pred_r.device
        sym = int((not refine) and int(idx.reshape(-1)[0].item()) in self.sym_list)
        loss, dis = torch.empty(1, device=dev), torch.empty(1, device=dev)
        new_points, new_target = torch.empty(1, N, 3, device=dev), torch.empty(1, M, 3, device=dev)
        scratch = torch.empty(N, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().df_loss_forward(pred_r.data_ptr(), pred_t.data_ptr(), pred_c.data_ptr(), target.data_ptr(),
                                            model_points.data_ptr(), points.data_ptr(), N, M, float(w), sym,
                                            loss.data_ptr(), dis.data_ptr(), new_points.data_ptr(), new_target.data_ptr(),
                                            scratch.data_ptr(), _lib.current_stream())
        _lib.check(st, "loss_forward")
        return loss[0], dis[0], new_points, new_target

    __call__ = forward
