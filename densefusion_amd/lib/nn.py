"""``nn_distance`` -- host-side mirror of lib/nn.py:3-35 on the fused 1-NN kernel.

``nn_distance(pc1 [B,N,C], pc2 [B,M,C]) -> (dist1 [B,N], idx1 [B,N], dist2 [B,M], idx2 [B,M])``: squared-L2 distance
to, and 0-based index of, the nearest point of the other cloud, both directions.  The reference materialises the
[B,N,M,C] difference tensor; here each direction is one ``df_knn`` launch (nothing of size N*M exists) and the
distances are evaluated only for the N + M matched pairs.  The ``l1smooth`` / ``l1`` variants are never used on the
DenseFusion path (lib/loss.py, lib/loss_refiner.py call it with the defaults) and are refused.
"""
from __future__ import annotations

import torch

from .knn import KNearestNeighbor

_knn1 = KNearestNeighbor(1)


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    if l1smooth or l1:
        raise NotImplementedError("nn_distance: only the squared-L2 distance (the defaults) is on the device path")
    if pc1.dim() != 3 or pc2.dim() != 3 or pc1.shape[0] != pc2.shape[0] or pc1.shape[2] != pc2.shape[2]:
        raise RuntimeError("nn_distance: expected pc1 [B,N,C] and pc2 [B,M,C]")
    pc1, pc2 = pc1.float().cuda(), pc2.float().cuda()
    a, b = pc1.transpose(2, 1).contiguous(), pc2.transpose(2, 1).contiguous()         # [B,C,N], [B,C,M]
    idx1 = _knn1(b, a)[:, 0] - 1                                                        # nearest pc2 point of every pc1 point
    idx2 = _knn1(a, b)[:, 0] - 1
    C = pc1.shape[2]
    dist1 = torch.sum((pc1 - torch.gather(pc2, 1, idx1.unsqueeze(-1).expand(-1, -1, C))) ** 2, dim=-1)
    dist2 = torch.sum((pc2 - torch.gather(pc1, 1, idx2.unsqueeze(-1).expand(-1, -1, C))) ** 2, dim=-1)
    assert not torch.any(torch.isnan(dist1))
    assert not torch.any(torch.isnan(dist2))
    return dist1, idx1, dist2, idx2
