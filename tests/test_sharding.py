"""CPU: the N>1 host path -- bucketed round-robin sharding + pose gather, 2 gloo ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from densefusion_amd import sharding

SIZES = [(80, 80), (120, 160), (80, 80), (160, 160), (120, 160), (80, 80), (240, 320), (160, 160), (80, 80)]


def test_shard_plan_partitions_every_object_once():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            for hw, idxs in sharding.shard_plan(SIZES, world, r).items():
                assert all(tuple(SIZES[i]) == hw for i in idxs)       # buckets are size-pure
                seen += idxs
        assert sorted(seen) == list(range(len(SIZES)))
    with pytest.raises(ValueError):
        sharding.shard_plan(SIZES, 2, 2)
    assert sharding.shard_plan([], 2, 0) == {}


def test_bucket_balance():
    sizes = [(80, 80)] * 16 + [(160, 160)] * 8
    loads = [sum(len(v) for v in sharding.shard_plan(sizes, 8, r).values()) for r in range(8)]
    assert max(loads) - min(loads) == 0


def _fake_pose(i):
    return torch.tensor([np.cos(i), np.sin(i), 0.0, 0.0, i, 2.0 * i, -i], dtype=torch.float64)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = sharding.shard_plan(SIZES, world, rank)
    idx = [i for v in plan.values() for i in v]
    poses = torch.stack([_fake_pose(i) for i in idx]) if idx else torch.zeros(0, 7, dtype=torch.float64)
    full = sharding.gather_poses(idx, poses, len(SIZES))
    want = torch.stack([_fake_pose(i) for i in range(len(SIZES))])
    ret[rank] = bool(torch.equal(full, want))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_poses_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))


def test_gather_poses_single_process():
    idx = [2, 0, 1]
    poses = torch.stack([_fake_pose(i) for i in idx])
    full = sharding.gather_poses(idx, poses, 3)
    assert torch.equal(full, torch.stack([_fake_pose(i) for i in range(3)]))
