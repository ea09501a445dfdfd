"""Training plumbing: flat parameter buffer + gradient all-reduce on 2 gloo ranks (CPU); Adam kernel on GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from densefusion_amd import train_utils


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(5, 7), nn.ReLU(), nn.Linear(7, 3))


def test_flat_params_alias_parameters_and_grads():
    m = _model()
    before = [p.detach().clone() for p in m.parameters()]
    flat = train_utils.FlatParams(m)
    assert flat.numel == sum(p.numel() for p in m.parameters())
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)
    m(torch.ones(2, 5)).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.equal(g, flat.grad) and float(flat.grad.abs().sum()) > 0          # grads landed in the flat buffer
    flat.data.mul_(0.5)                                                              # updating the buffer updates the model
    for p, b in zip(m.parameters(), before):
        assert torch.allclose(p, b * 0.5)
    flat.zero_grad()
    assert all(float(p.grad.abs().sum()) == 0 for p in m.parameters())


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    flat = train_utils.FlatParams(m)
    x = torch.full((1, 5), float(rank + 1))
    m(x).sum().backward()
    local = flat.grad.clone()
    n = train_utils.allreduce_gradients(flat)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ret[rank] = bool(n == world and torch.allclose(flat.grad, sum(gathered)))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def test_allreduce_without_process_group_is_identity():
    m = _model()
    flat = train_utils.FlatParams(m)
    m(torch.ones(1, 5)).sum().backward()
    g = flat.grad.clone()
    assert train_utils.allreduce_gradients(flat) == 1 and torch.equal(flat.grad, g)


@pytest.mark.gpu
def test_flat_adam_matches_torch_optim_adam():
    ref = _model()
    mine = _model().cuda()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    flat = train_utils.FlatParams(mine)
    adam = train_utils.FlatAdam(flat, lr=1e-3)
    for it in range(5):
        x = torch.randn(4, 5, generator=torch.Generator().manual_seed(it))
        opt.zero_grad(); ref(x).pow(2).sum().backward(); opt.step()
        flat.zero_grad(); mine(x.cuda()).pow(2).sum().backward(); adam.step()
    for a, b in zip(mine.parameters(), ref.parameters()):
        assert torch.allclose(a.cpu(), b, rtol=1e-5, atol=1e-6)
    # grad_scale = averaging factor of the data-parallel exchange
    flat.zero_grad(); mine(torch.ones(4, 5).cuda()).sum().backward()
    flat.grad.mul_(4.0)
    p0 = flat.data.clone()
    adam.step(grad_scale=0.25)
    assert not torch.equal(p0, flat.data)


def _sync_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # what tools/train.py does: per-rank seeds -> per-rank initial weights
    a, b = nn.Linear(4, 3), nn.Linear(3, 2)
    apart = train_utils.replicas_in_sync([a, b])
    for prm in a.parameters():
        dist.broadcast(prm.data, 0)
    half = train_utils.replicas_in_sync([a, b])          # only one of the two networks was synchronised
    for prm in b.parameters():
        dist.broadcast(prm.data, 0)
    ret[rank] = (apart, half, train_utils.replicas_in_sync([a, b]))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_in_sync_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert dict(ret) == {0: (False, False, True), 1: (False, False, True)}
