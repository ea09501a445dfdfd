"""Dev tool: time a list of per-point GEMM shapes with the 128x128 and the 64x64 tile (DF_IGEMM_TILE read per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd.ops import conv2d_nhwc

SHAPES = [(192000, 256, 576), (40960, 192, 512), (40960, 384, 640), (40960, 640, 256), (40960, 256, 128), (40960, 256, 512),
          (48000, 512, 1024), (40960, 512, 1024), (40960, 576, 64), (12800, 256, 256), (12000, 256, 512)]

def run(M, K, N, tile):
    if tile: os.environ["DF_IGEMM_TILE"] = tile
    else: os.environ.pop("DF_IGEMM_TILE", None)
    x = torch.randn(M, 1, 1, K, device="cuda"); w = torch.randn(N, 1, 1, K, device="cuda") * 0.02
    for _ in range(3): conv2d_nhwc(x, w, act=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): conv2d_nhwc(x, w, act=1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    return us, 2.0 * M * K * N / us / 1e6

for (M, K, N) in SHAPES:
    a = run(M, K, N, "a"); c = run(M, K, N, "c"); d = run(M, K, N, None)
    extra = ""
    if len(sys.argv) > 1:                                  # name of an env switch to flip for a third column
        os.environ[sys.argv[1]] = "1"
        e = run(M, K, N, None)
        os.environ.pop(sys.argv[1])
        extra = f" | {sys.argv[1]}=1 {e[0]:7.1f} us {e[1]:6.1f} TF"
    print(f"M={M} K={K} N={N}: 128x128 {a[0]:7.1f} us {a[1]:6.1f} TF | 64x64 {c[0]:7.1f} us {c[1]:6.1f} TF | model {d[0]:7.1f} us {d[1]:6.1f} TF{extra}")
