"""Micro-benchmark of the fused 1-NN kernel (dev tool): prints per-launch time and roofline fractions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd.lib.knn import KNearestNeighbor

def main():
    knn = KNearestNeighbor(1)
    sizes = [(1, 500, 500000), (1, 500, 1000000), (8, 500, 1000000), (1, 2600, 2600), (1, 500, 500)]
    if len(sys.argv) == 4:                                     # one size only (for rocprofv3 --pmc passes)
        sizes = [tuple(int(a) for a in sys.argv[1:4])]
    for (B, R, Q) in sizes:
        ref = (torch.rand(B, 3, R, device="cuda") - 0.5) * 0.2
        qry = (torch.rand(B, 3, Q, device="cuda") - 0.5) * 0.25
        for _ in range(3):
            knn(ref, qry)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            knn(ref, qry)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        byts = B * (12 * (R + Q) + 8 * Q)
        flops = 9.0 * B * R * Q
        print(f"B={B} R={R} Q={Q}: {ms*1e3:.1f} us  {byts/ms/1e6:.1f} GB/s ({byts/ms/1e6/8000*100:.2f}% of 8 TB/s)  "
              f"{flops/ms/1e9:.2f} TFLOP/s ({flops/ms/1e9/157.3*100:.1f}% of fp32 peak)  {B*R*Q/ms/1e6:.1f} Gpairs/s")

if __name__ == "__main__":
    main()
