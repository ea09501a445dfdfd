"""ADD / ADD-S metric on the device (tools/eval_linemod.py:118-130), batched over objects."""
from __future__ import annotations

import torch

from .. import _lib


def add_metric(pose, model_points, target, symmetric=None):
    """pose [B,7] fp64 (q wxyz, t); model_points, target [B,M,3] fp32; symmetric [B] bool/int or None.
    Returns [B] fp64: mean ||R m + t - target|| (ADD) or mean distance to the nearest target point (ADD-S)."""
    pose = pose.detach().double().contiguous()
    mp, tg = model_points.detach().float().contiguous(), target.detach().float().contiguous()
    if not (pose.is_cuda and mp.is_cuda and tg.is_cuda):
        raise RuntimeError("densefusion_amd needs device tensors (no CPU path)")
    B, M = mp.shape[0], mp.shape[1]
    out = torch.empty(B, dtype=torch.float64, device=pose.device)
    sym = None
    if symmetric is not None:
        sym = torch.as_tensor(symmetric).to(device=pose.device, dtype=torch.int32).contiguous()
    with _lib.device_guard(pose.device):
        st = _lib.lib().df_add_metric(pose.data_ptr(), mp.data_ptr(), tg.data_ptr(), sym.data_ptr() if sym is not None else None,
                                      B, M, out.data_ptr(), _lib.current_stream())
    _lib.check(st, "add_metric")
    return out
