"""Dev tool: A/B the implicit-GEMM kernel variants on the network's big layer shapes, interleaved in one
process (v1 = un-pipelined reference kernel via DF_IGEMM_V1, v2 = software-pipelined)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd.ops import conv2d_nhwc

SHAPES = [  # B,H,W,Cin,Cout,k,pad,dil
    (10, 40, 40, 1024, 256, 3, 1, 1),    # up_1 @160x160 x10
    (10, 20, 20, 512, 512, 3, 4, 4),     # layer4.1
    (10, 20, 20, 2560, 1024, 1, 0, 1),   # psp bottleneck
    (10, 80, 80, 256, 64, 3, 1, 1),      # up_2
    (10, 160, 160, 64, 64, 3, 1, 1),     # up_3
    (10240, 1, 1, 384, 1920, 1, 0, 1),   # head layer 1
    (10240, 1, 1, 512, 1024, 1, 0, 1),   # conv6
    (10, 20, 20, 256, 256, 3, 2, 2),     # layer3.1
    (10, 40, 40, 64, 64, 3, 1, 1),       # layer1
    (10, 20, 20, 1024, 2304, 1, 0, 1),   # up_1 as low-res per-tap GEMM
    (10, 40, 40, 256, 576, 1, 0, 1),     # up_2 low-res
    (10, 80, 80, 64, 576, 1, 0, 1),      # up_3 low-res
    (10, 20, 20, 128, 128, 3, 1, 1),     # layer2
    (10, 10, 10, 512, 512, 3, 4, 4),     # layer4.1 @80x80
]

def bench(x, w, pad, dil, n=10):
    for _ in range(2):
        conv2d_nhwc(x, w, pad=pad, dil=dil, act=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        conv2d_nhwc(x, w, pad=pad, dil=dil, act=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

VA = {"DF_IGEMM_TILE": os.environ.get("TILE_A", "c")}   # variant A env
VB = {"DF_IGEMM_TILE": os.environ.get("TILE_B", "b")}   # variant B env


def main():
    print("A =", VA, " B =", VB)
    for (B, H, W, Cin, Cout, k, pad, dil) in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.02
        fl = 2.0 * B * H * W * Cout * k * k * Cin
        res = {}
        variants = [("A", VA), ("B", VB)]
        for rnd in range(3):
            for name, env in variants:
                for kk in ("DF_IGEMM_V1", "DF_IGEMM_TILE"):
                    os.environ.pop(kk, None)
                os.environ.update(env)
                res.setdefault(name, []).append(bench(x, w, pad, dil))
        for kk in ("DF_IGEMM_V1", "DF_IGEMM_TILE"):
            os.environ.pop(kk, None)
        v1, v2 = min(res["A"]), min(res["B"])
        print(f"M={B*H*W:6d} N={Cout:4d} K={k*k*Cin:5d}: A {v1*1e3:7.1f} us {fl/v1/1e9:6.1f} TF | B {v2*1e3:7.1f} us {fl/v2/1e9:6.1f} TF | x{v1/v2:.3f}")

if __name__ == "__main__":
    main()
