"""Data-parallel sharding of (frame, object) units over the GPUs of one node.

Every pose is independent (the object loop of tools/eval_ycb.py:147 carries no cross-object state), so
the path shards by objects: one process per GPU, weights replicated, NO collective on the data path.
Objects are first bucketed by their snapped crop size (multiples of 40 px,
datasets/ycb/dataset.py:247-289) because one batched launch sequence needs same-size crops, then every
bucket is dealt round-robin over the ranks.  The only communication is one all_gather of the [n,7]
poses (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from collections import OrderedDict

import torch


def bucket_by_size(sizes):
    """sizes: sequence of (H, W) per object -> OrderedDict {(H, W): [object indices]} (stable order)."""
    out = OrderedDict()
    for i, hw in enumerate(sizes):
        out.setdefault((int(hw[0]), int(hw[1])), []).append(i)
    return out


def shard_plan(sizes, world_size: int, rank: int):
    """The objects rank `rank` evaluates: {(H, W): [indices]}, each bucket dealt round-robin."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    plan = OrderedDict()
    for hw, idxs in bucket_by_size(sizes).items():
        mine = idxs[rank::world_size]
        if mine:
            plan[hw] = mine
    return plan


def gather_poses(local_idx, local_poses, total: int, group=None):
    """All ranks contribute (indices [n_i], poses [n_i,7]); every rank returns the full [total,7] table.

    Ragged shards are padded to the largest shard so one all_gather suffices."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = local_poses.device
    idx = torch.as_tensor(local_idx, dtype=torch.int64, device=dev)
    if world == 1:
        full = torch.zeros(total, 7, dtype=local_poses.dtype, device=dev)
        full[idx] = local_poses
        return full
    n = torch.tensor([idx.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    cap = int(max(int(c) for c in counts))
    pad_idx = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    pad_idx[:idx.numel()] = idx
    pad_pose = torch.zeros(cap, 7, dtype=local_poses.dtype, device=dev)
    pad_pose[:idx.numel()] = local_poses
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_pose = [torch.empty_like(pad_pose) for _ in range(world)]
    dist.all_gather(all_idx, pad_idx, group=group)
    dist.all_gather(all_pose, pad_pose, group=group)
    full = torch.zeros(total, 7, dtype=local_poses.dtype, device=dev)
    for i, p in zip(all_idx, all_pose):
        keep = i >= 0
        full[i[keep]] = p[keep]
    return full
