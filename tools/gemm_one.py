"""Dev tool: run one conv shape a few times (for rocprofv3 --pmc runs). args: B H W Cin Cout k pad dil"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd.ops import conv2d_nhwc
B, H, W, Cin, Cout, k, pad, dil = [int(v) for v in sys.argv[1:9]]
x = torch.randn(B, H, W, Cin, device="cuda")
w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.02
for _ in range(5):
    conv2d_nhwc(x, w, pad=pad, dil=dil, act=1)
torch.cuda.synchronize()
