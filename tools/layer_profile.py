"""Dev tool: per-GEMM-launch time / TFLOP/s of one batched bucket (DF_PROFILE_VERBOSE dump)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DF_DEV_LIB"] = "1"          # the development build: the only one that reads switches
os.environ["DF_PROFILE_VERBOSE"] = "1"
import torch
import bench
from densefusion_amd import _lib

def main():
    H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    bench.CROPS[:] = [(H, W)]
    dev = torch.device("cuda", 0)
    est, ref = bench.load_nets(dev)
    pe = [bench.PoseEstimator(est, ref)]
    buckets = bench.make_groups(bench.make_buckets(0, 1, B, dev), 1, dev)
    for _ in range(3):
        bench.run_step(pe, buckets)
    torch.cuda.synchronize()
    ms, fl, useful, by, n = bench.profile_gemm(pe, buckets, 1)
    print(f"total gemm {ms:.3f} ms, {fl/ms/1e9:.1f} TFLOP/s, {n} launches")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        bench.run_step(pe, buckets)
    e1.record(); torch.cuda.synchronize()
    print(f"step (eager) {e0.elapsed_time(e1)/5:.3f} ms for {B} poses of {H}x{W}")

if __name__ == "__main__":
    main()
