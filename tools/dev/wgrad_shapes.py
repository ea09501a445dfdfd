"""Dev tool: weight-gradient launches of the YCB training step (8 frames per pass, 160x160 crop) one by one: us, TFLOP/s."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd.ops import _desc, wgrad
SHAPES = [  # B, H, W, Cin, Cout, k, pad, dil, stride
    (8, 160, 160, 4, 64, 7, 3, 1, 2), (8, 40, 40, 64, 64, 3, 1, 1, 1), (8, 40, 40, 64, 128, 3, 1, 1, 2), (8, 20, 20, 128, 128, 3, 1, 1, 1),
    (8, 20, 20, 128, 256, 3, 1, 1, 1), (8, 20, 20, 256, 256, 3, 2, 2, 1), (8, 20, 20, 256, 512, 3, 1, 1, 1), (8, 20, 20, 512, 512, 3, 4, 4, 1),
    (8, 20, 20, 2560, 1024, 1, 0, 1, 1), (8, 40, 40, 1024, 256, 3, 1, 1, 1), (8, 80, 80, 256, 64, 3, 1, 1, 1), (8, 160, 160, 64, 64, 3, 1, 1, 1),
    (8, 160, 160, 64, 32, 1, 0, 1, 1), (8, 1000, 1, 512, 1024, 1, 0, 1, 1), (8, 1000, 1, 1408, 640, 1, 0, 1, 1), (8, 1000, 1, 640, 256, 1, 0, 1, 1),
    (8, 1000, 1, 64, 128, 1, 0, 1, 1), (8, 1000, 1, 256, 512, 1, 0, 1, 1)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
tot_us = tot_fl = 0
for (B, H, W, Cin, Cout, k, pad, dil, stride) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, k, k, Cin, device="cuda")
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = torch.randn(B, OH, OW, Cout, device="cuda")
    dw, db = torch.empty_like(w), torch.empty(Cout, device="cuda")
    d = _desc(x, w, None, stride, pad, dil)
    for _ in range(3):
        wgrad(d, dy, dw, db)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        wgrad(d, dy, dw, db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * B * OH * OW * Cout * k * k * Cin
    tot_us += us; tot_fl += fl
    print(f"M={B*OH*OW:7d} Cout={Cout:5d} K={k*k*Cin:6d}  {us:8.1f} us  {fl/us/1e6:6.1f} TF")
print(f"total {tot_us:.0f} us, {tot_fl/tot_us/1e6:.1f} TF")
