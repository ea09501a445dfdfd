"""``KNearestNeighbor`` -- host-side mirror of the reference operator (lib/knn/__init__.py:9-23).

Same call shape: ``KNearestNeighbor(k)(ref[B,D,R], query[B,D,Q]) -> int64 [B,k,Q]`` with 1-based
indices of the k nearest reference points per query column.  The work is done by the fused HIP
kernel behind ``df_knn`` (include/dfusion.h); the reference's R*Q distance scratch is never allocated.
"""
from __future__ import annotations

import torch

from ... import _lib


class KNearestNeighbor:
    def __init__(self, k):
        self.k = int(k)

    def forward(self, ref, query):
        # reference: ref.float().cuda(), query.float().cuda() (lib/knn/__init__.py:16-17)
        ref = ref.float().cuda().contiguous()
        query = query.float().cuda().contiguous()
        if ref.dim() != 3:
            raise RuntimeError("ref_tensor: 3D Tensor expected")          # knn_pytorch.c:11
        if query.dim() != 3:
            raise RuntimeError("query_tensor: 3D Tensor expected")        # knn_pytorch.c:12
        if ref.shape[0] != query.shape[0] or ref.shape[1] != query.shape[1]:
            raise RuntimeError("input sizes must match")                  # knn_pytorch.c:14-15
        B, D, R = ref.shape
        Q = query.shape[2]
        inds = torch.empty(B, self.k, Q, dtype=torch.int64, device=query.device)
        with _lib.device_guard(query.device):
            st = _lib.lib().df_knn(_lib.dptr(ref), _lib.dptr(query), _lib.dptr(inds), B, D, R, Q, self.k,
                                   _lib.current_stream())
        _lib.check(st, "knn")
        return inds

    __call__ = forward
