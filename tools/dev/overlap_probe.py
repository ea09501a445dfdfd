"""Dev tool: does a memory-bound kernel on one stream hide under the matrix-core GEMM on another?  Times a plain GEMM
(131072 x 1024 x 1024: the 128x128-tile kernel, 4 workgroups per CU) and an elementwise pass over 1 GB alone and side by side,
on two streams tested to sit on different hardware queues.  usage: overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import ops
from densefusion_amd.streams import concurrent_streams

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
M, K, N = 131072, 1024, 1024
x = torch.randn(1, M, 1, K, device=dev)
w = torch.randn(N, 1, 1, K, device=dev) * 0.03
out = torch.empty(1, M, 1, N, device=dev)
a = torch.randn(256 << 20, device=dev)          # 1 GB
b = torch.empty_like(a)
sa, sb = concurrent_streams(dev, 2)
REP_G, REP_E = 8, 24


def gemm():
    with torch.cuda.stream(sa):
        for _ in range(REP_G):
            ops.conv2d_nhwc(x, w, out=out)


def elem():
    with torch.cuda.stream(sb):
        for _ in range(REP_E):
            torch.mul(a, 1.0001, out=b)


def timed(*fs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fs:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for f in (gemm, elem):
    f()
tg = min(timed(gemm) for _ in range(3))
te = min(timed(elem) for _ in range(3))
tb = min(timed(gemm, elem) for _ in range(3))
fl = 2.0 * M * K * N * REP_G
print(f"GEMM alone {tg:.2f} ms ({fl / tg / 1e9:.1f} TFLOP/s), elementwise alone {te:.2f} ms ({REP_E * 2.0 * a.numel() * 4 / te / 1e6:.0f} GB/s), "
      f"both {tb:.2f} ms (sum {tg + te:.2f}, max {max(tg, te):.2f})")
