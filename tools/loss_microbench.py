"""Dev tool: the symmetric PoseNet loss forward (csrc/loss.hip add_dis_sym_kernel: transform + 1-NN + reduction in one launch) at
BASELINE configs[3]'s size, N = 1000 poses x M = 500 mesh points = 250 M pairs; for rocprofv3 passes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd import _lib

def main():
    N, M, dev = 1000, 500, "cuda"
    g = torch.Generator(device="cpu").manual_seed(7)
    q = torch.randn(N, 4, generator=g).to(dev); pt = (torch.randn(N, 3, generator=g) * 0.03).to(dev)
    pc = (torch.rand(N, generator=g) * 0.9 + 0.05).to(dev)
    mp = ((torch.rand(M, 3, generator=g) - 0.5) * 0.2).to(dev); tgt = mp + 0.5; pts = (torch.rand(N, 3, generator=g) * 0.1 + 0.45).to(dev)
    loss, dis = torch.empty(1, device=dev), torch.empty(1, device=dev)
    npts, ntgt, scratch = torch.empty(N, 3, device=dev), torch.empty(M, 3, device=dev), torch.empty(N, device=dev)
    sel = torch.empty(N, M, dtype=torch.int32, device=dev)
    L = _lib.lib()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(23):
        if it == 3:
            torch.cuda.synchronize(); e0.record()
        _lib.check(L.df_loss_forward(q.data_ptr(), pt.data_ptr(), pc.data_ptr(), tgt.data_ptr(), mp.data_ptr(), pts.data_ptr(), N, M,
                                     ctypes.c_float(0.015), 1, loss.data_ptr(), dis.data_ptr(), npts.data_ptr(), ntgt.data_ptr(),
                                     scratch.data_ptr(), sel.data_ptr(), _lib.current_stream()), "loss_forward")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"symmetric loss forward N={N} M={M}: {us:.1f} us per call, {9.0*N*M*M/us/1e6:.2f} TFLOP/s")

if __name__ == "__main__":
    main()
