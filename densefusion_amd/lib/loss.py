"""``Loss`` -- host-side mirror of lib/loss.py:73-82 over the fused HIP kernels, forward AND backward.

``Loss(num_points_mesh, sym_list).forward(pred_r, pred_t, pred_c, target, model_points, idx, points, w,
refine)`` -> ``(loss, dis, new_points [1,N,3], new_target [1,M,3])`` with the reference's shapes.  ``loss``
carries a grad_fn (``loss.backward()`` works, tools/train.py:161): the backward is one more fused kernel
(``df_loss_backward``) that re-derives the transformed points and chains through the quaternion map; the
nearest-neighbour match of the symmetric branch is saved by the forward pass and enters as a constant, as
``torch.index_select`` does in the reference.  ``dis``, ``new_points`` and ``new_target`` are detached, as in
the reference (lib/loss.py:70).

The symmetric branch implements the semantics the reference intends (lib/knn 1-NN of every transformed
model point among the target points, lib/loss.py:9,41-47); in this fork that branch raises because of a
mis-wired import (SURVEY header note 3).
"""
from __future__ import annotations

import torch

from .. import _lib
from ..train_utils import host_index


def _f32(t):
    if not t.is_cuda:
        raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
    return t.detach().float().contiguous()


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_r, pred_t, pred_c, target, model_points, points, w, sym, M):
        pr, pt, pc = _f32(pred_r), _f32(pred_t), _f32(pred_c)
        tg, mp, pts = _f32(target), _f32(model_points), _f32(points)
        N, dev = pc.shape[1], pr.device
        loss, dis = torch.empty(1, device=dev), torch.empty(1, device=dev)
        new_points, new_target = torch.empty(1, N, 3, device=dev), torch.empty(1, M, 3, device=dev)
        scratch = torch.empty(N, device=dev)
        sel = torch.empty(N, M, dtype=torch.int32, device=dev) if sym else None
        with _lib.device_guard(dev):
            st = _lib.lib().df_loss_forward(pr.data_ptr(), pt.data_ptr(), pc.data_ptr(), tg.data_ptr(), mp.data_ptr(),
                                            pts.data_ptr(), N, M, float(w), int(sym), loss.data_ptr(), dis.data_ptr(),
                                            new_points.data_ptr(), new_target.data_ptr(), scratch.data_ptr(),
                                            sel.data_ptr() if sel is not None else None, _lib.current_stream())
        _lib.check(st, "loss_forward")
        ctx.save_for_backward(pr, pt, pc, tg, mp, pts, scratch, sel if sel is not None else torch.empty(0, device=dev))
        ctx.meta = (N, M, float(w), bool(sym))
        ctx.mark_non_differentiable(dis, new_points, new_target)
        return loss[0], dis[0], new_points, new_target

    @staticmethod
    def backward(ctx, g_loss, _g_dis, _g_np, _g_nt):
        pr, pt, pc, tg, mp, pts, dis, sel = ctx.saved_tensors
        N, M, w, sym = ctx.meta
        d_r, d_t, d_c = torch.empty_like(pr), torch.empty_like(pt), torch.empty_like(pc)
        with _lib.device_guard(pr.device):
            st = _lib.lib().df_loss_backward(pr.data_ptr(), pt.data_ptr(), pc.data_ptr(), tg.data_ptr(), mp.data_ptr(),
                                             pts.data_ptr(), sel.data_ptr() if sym else None, dis.data_ptr(), N, M, w,
                                             float(g_loss), d_r.data_ptr(), d_t.data_ptr(), d_c.data_ptr(), _lib.current_stream())
        _lib.check(st, "loss_backward")
        return d_r, d_t, d_c, None, None, None, None, None, None


class Loss:
    def __init__(self, num_points_mesh, sym_list):
        self.num_pt_mesh = int(num_points_mesh)
        self.sym_list = list(sym_list)

    def forward(self, pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine):
        bs, N = pred_c.shape[0], pred_c.shape[1]
        M = self.num_pt_mesh
        if bs != 1 or target.numel() != M * 3 or model_points.numel() != M * 3 or points.numel() != N * 3:
            raise RuntimeError("Loss.forward: expected bs = 1, target/model_points [1,M,3], points [1,N,3]")
        if not pred_r.is_cuda:
            raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
        sym = (not refine) and host_index(idx) in self.sym_list
        return _LossFn.apply(pred_r, pred_t, pred_c, target, model_points, points, w, sym, M)

    __call__ = forward

    def forward_frames(self, pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine):
        """``forward`` for B stacked frames in a handful of launches (``df_loss_forward_frames``), WITHOUT an autograd graph: pred_r [B,N,4],
        pred_t [B,N,3], pred_c [B,N,1]|[B,N], target / model_points [B,M,3], points [B,N,3]; ``idx``: host object indices (an int per
        frame) -> (loss [B], dis [B], new_points [B,N,3], new_target [B,M,3]).  Each frame's numbers are those of a ``forward`` call on it.
        (The refiner phase of tools/train.py asks for the whole window's re-centred points at once.)"""
        import ctypes
        pr, pt, pc, tg, mp, pts = (_f32(t) for t in (pred_r, pred_t, pred_c, target, model_points, points))
        B, N, M, dev = pr.shape[0], pr.shape[1], self.num_pt_mesh, pr.device
        if pr.shape != (B, N, 4) or pt.shape != (B, N, 3) or pc.numel() != B * N or tg.shape != (B, M, 3) or mp.shape != (B, M, 3) or pts.shape != (B, N, 3):
            raise RuntimeError("Loss.forward_frames: expected pred_r [B,N,4], pred_t [B,N,3], pred_c [B,N], target / model_points [B,M,3], points [B,N,3]")
        sym = [int((not refine) and int(i) in self.sym_list) for i in idx]
        if len(sym) != B:
            raise RuntimeError("Loss.forward_frames: one object index per frame")
        loss, dis = torch.empty(B, device=dev), torch.empty(B, device=dev)
        new_points, new_target = torch.empty(B, N, 3, device=dev), torch.empty(B, M, 3, device=dev)
        scratch = torch.empty(B, N, device=dev)
        sel = torch.empty(B, N, M, dtype=torch.int32, device=dev) if any(sym) else None
        with _lib.device_guard(dev):
            st = _lib.lib().df_loss_forward_frames(B, (ctypes.c_int * B)(*sym), pr.data_ptr(), pt.data_ptr(), pc.data_ptr(), tg.data_ptr(), mp.data_ptr(),
                                                   pts.data_ptr(), N, M, float(w), loss.data_ptr(), dis.data_ptr(), new_points.data_ptr(),
                                                   new_target.data_ptr(), scratch.data_ptr(), sel.data_ptr() if sel is not None else None,
                                                   _lib.current_stream())
        _lib.check(st, "loss_forward_frames")
        return loss, dis, new_points, new_target
