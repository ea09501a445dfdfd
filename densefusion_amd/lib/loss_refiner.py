"""``Loss_refine`` -- host-side mirror of lib/loss_refiner.py:65-74 over the fused HIP kernels (forward + backward).

``forward(pred_r [1,4], pred_t [1,3], target [1,M,3], model_points [1,M,3], idx, points [1,N,3])`` ->
``(dis [1], new_points [1,N,3], new_target [1,M,3])``; ``dis.backward()`` works (tools/train.py:159).
Symmetric objects take the 1-NN (ADD-S) branch with the intended lib/knn semantics (lib/loss_refiner.py:40-46).
"""
from __future__ import annotations

import torch

from .. import _lib
from .loss import _f32
from ..train_utils import host_index


class _LossRefineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_r, pred_t, target, model_points, points, sym, M):
        pr, pt = _f32(pred_r).view(-1), _f32(pred_t).view(-1)
        tg, mp, pts = _f32(target), _f32(model_points), _f32(points)
        N, dev = pts.shape[1], pr.device
        dis = torch.empty(1, device=dev)
        new_points, new_target = torch.empty(1, N, 3, device=dev), torch.empty(1, M, 3, device=dev)
        sel = torch.empty(M, dtype=torch.int32, device=dev) if sym else None
        with _lib.device_guard(dev):
            st = _lib.lib().df_loss_refine_forward(pr.data_ptr(), pt.data_ptr(), tg.data_ptr(), mp.data_ptr(), pts.data_ptr(), N, M,
                                                   int(sym), dis.data_ptr(), new_points.data_ptr(), new_target.data_ptr(),
                                                   sel.data_ptr() if sel is not None else None, _lib.current_stream())
        _lib.check(st, "loss_refine_forward")
        ctx.save_for_backward(pr, pt, tg, mp, sel if sel is not None else torch.empty(0, device=dev))
        ctx.meta = (M, bool(sym), pred_r.shape, pred_t.shape)
        ctx.mark_non_differentiable(new_points, new_target)
        return dis, new_points, new_target

    @staticmethod
    def backward(ctx, g_dis, _g_np, _g_nt):
        pr, pt, tg, mp, sel = ctx.saved_tensors
        M, sym, shp_r, shp_t = ctx.meta
        d_r, d_t = torch.empty_like(pr), torch.empty_like(pt)
        with _lib.device_guard(pr.device):
            st = _lib.lib().df_loss_refine_backward(pr.data_ptr(), pt.data_ptr(), tg.data_ptr(), mp.data_ptr(),
                                                    sel.data_ptr() if sym else None, M, float(g_dis.reshape(-1)[0]),
                                                    d_r.data_ptr(), d_t.data_ptr(), _lib.current_stream())
        _lib.check(st, "loss_refine_backward")
        return d_r.view(shp_r), d_t.view(shp_t), None, None, None, None, None


class Loss_refine:
    def __init__(self, num_points_mesh, sym_list):
        self.num_pt_mesh = int(num_points_mesh)
        self.sym_list = list(sym_list)

    def forward(self, pred_r, pred_t, target, model_points, idx, points):
        M = self.num_pt_mesh
        if pred_r.numel() != 4 or pred_t.numel() != 3 or target.numel() != M * 3 or model_points.numel() != M * 3:
            raise RuntimeError("Loss_refine.forward: expected pred_r [1,4], pred_t [1,3], target/model_points [1,M,3]")
        if not pred_r.is_cuda:
            raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
        sym = host_index(idx) in self.sym_list
        return _LossRefineFn.apply(pred_r, pred_t, target, model_points, points, sym, M)

    __call__ = forward
