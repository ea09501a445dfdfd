"""``Loss_refine`` -- host-side mirror of lib/loss_refiner.py:65-74 over the fused HIP kernels (forward).

``forward(pred_r [1,4], pred_t [1,3], target [1,M,3], model_points [1,M,3], idx, points [1,N,3])`` ->
``(dis [1], new_points [1,N,3], new_target [1,M,3])``.  Symmetric objects take the 1-NN (ADD-S) branch
with the intended lib/knn semantics (lib/loss_refiner.py:40-46).
"""
from __future__ import annotations

import torch

from .. import _lib
from .loss import _f32


class Loss_refine:
    def __init__(self, num_points_mesh, sym_list):
        self.num_pt_mesh = int(num_points_mesh)
        self.sym_list = list(sym_list)

    def forward(self, pred_r, pred_t, target, model_points, idx, points):
        pred_r, pred_t = _f32(pred_r).view(-1), _f32(pred_t).view(-1)
        target, model_points, points = _f32(target), _f32(model_points), _f32(points)
        M = self.num_pt_mesh
        N = points.shape[1]
        if pred_r.numel() != 4 or pred_t.numel() != 3 or target.numel() != M * 3 or model_points.numel() != M * 3:
            raise RuntimeError("Loss_refine.forward: expected pred_r [1,4], pred_t [1,3], target/model_points [1,M,3]")
        dev = pred_r.device
        sym = int(int(idx.reshape(-1)[0].item()) in self.sym_list)
        dis = torch.empty(1, device=dev)
        new_points, new_target = torch.empty(1, N, 3, device=dev), torch.empty(1, M, 3, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().df_loss_refine_forward(pred_r.data_ptr(), pred_t.data_ptr(), target.data_ptr(),
                                                   model_points.data_ptr(), points.data_ptr(), N, M, sym, dis.data_ptr(),
                                                   new_points.data_ptr(), new_target.data_ptr(), _lib.current_stream())
        _lib.check(st, "loss_refine_forward")
        return dis, new_points, new_target

    __call__ = forward
