"""Worker of tests/test_rccl_gpu.py: RCCL ("nccl" backend) with world_size = 1 on the one GPU of the box -- library load,
stream semantics next to a replaying hipGraph with side streams, the inference path's all_gather and the training path's
all_reduce of the 85.8 MB flat gradient buffer.  Prints RCCL_OK on success."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from densefusion_amd import train_utils  # noqa: E402
from densefusion_amd.lib.network import PoseEstimator, PoseNet  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    est, ref = bench.load_nets(dev)
    buckets = bench.make_buckets(0, 1, 2, dev)                      # 2 objects per crop size
    groups = bench.make_groups(buckets, 2, dev)
    pe = [PoseEstimator(est, ref) for _ in groups]
    streams = [torch.cuda.Stream() for _ in groups]
    bench.run_step(pe, groups)                                      # eager: uploads weights, sizes the workspaces
    torch.cuda.synchronize()
    eager = torch.cat([g["out"][1] for g in groups]).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bench.run_step(pe, groups, streams)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        bench.run_step(pe, groups, streams)
    # the training path's collective: the flat PoseNet gradient buffer (21 449 056 floats = 85.8 MB), summed while the graph replays
    flat = train_utils.FlatParams(PoseNet(bench.N_PTS, bench.K_OBJ).to(dev))
    assert flat.numel == 21449056
    flat.grad.copy_(torch.arange(flat.numel, device=dev, dtype=torch.float32) % 1024)
    want = flat.grad.clone()
    gathered = [torch.empty(eager.shape, dtype=torch.float64, device=dev)]
    for _ in range(3):
        graph.replay()
        mine = torch.cat([g["out"][1] for g in groups])
        dist.all_gather(gathered, mine)                                # bench.py's gather()
        assert train_utils.allreduce_gradients(flat) == 1              # world 1: the helper skips the collective ...
        dist.all_reduce(flat.grad)                                     # ... so issue it unconditionally here
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(gathered[0], eager), "graph replay + all_gather differ from the eager poses"
    assert torch.equal(flat.grad, want) and float(t) == 1.5
    dist.destroy_process_group()
    print("RCCL_OK")


if __name__ == "__main__":
    main()
