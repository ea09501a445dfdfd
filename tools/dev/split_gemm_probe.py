"""Dev tool (GPU box, development library): the bf16 x 6 split-precision GEMM experiment (csrc/split_gemm.hip) against the fp32-MFMA kernel on the
bench step's dominant plain-GEMM shapes: error of each against an fp64 product of the same fp32 operands, and TFLOP/s (fp32-equivalent: 2 M N K).
Run twice -- the switch is read once per process:
    DF_DEV_LIB=1 python tools/dev/split_gemm_probe.py                       (fp32 MFMA)
    DF_DEV_LIB=1 DF_GEMM_SPLIT_BF16=1 python tools/dev/split_gemm_probe.py  (bf16 x 6)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import ops

assert os.environ.get("DF_DEV_LIB"), "needs the development library (DF_DEV_LIB=1)"
mode = "bf16x6" if os.environ.get("DF_GEMM_SPLIT_BF16") else "fp32_mfma"
dev = torch.device("cuda")
# (name, M, N, K): M = pixels / points of the 280-object bench step
SHAPES = [("psp bottleneck (1/8 res), K=1024->2304 stacked", 139000, 2304, 1024), ("per-point 512->1024", 286720, 1024, 512),
          ("up_1 1x1 part 1024->512 @ 1/8", 139000, 1024, 512), ("per-point 640->256", 286720, 256, 640), ("per-point 256->512", 286720, 512, 256),
          ("small-M tail", 1000, 512, 512), ("M not a multiple of 128", 12345, 256, 192)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
out = {"mode": mode, "shapes": []}
keep = []          # the weight planes are cached per (pointer, size): no weight buffer may be recycled inside this process
for name, M, N, K in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).abs_().to(dev)                 # post-ReLU-like activations
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    y = ops.conv2d_nhwc(x.view(1, M, 1, K), w.view(N, 1, 1, K), bias=b, act=1).view(M, N)
    rows = torch.arange(0, M, max(1, M // 4096), device=dev)           # a sample of rows for the fp64 check (all columns)
    ref = torch.relu(x[rows].double() @ w.double().t() + b.double())
    err = (y[rows].double() - ref).abs()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    outbuf = torch.empty(1, M, 1, N, device=dev)
    for _ in range(2):
        ops.conv2d_nhwc(x.view(1, M, 1, K), w.view(N, 1, 1, K), bias=b, act=1, out=outbuf)
    e0.record()
    reps = 10
    for _ in range(reps):
        ops.conv2d_nhwc(x.view(1, M, 1, K), w.view(N, 1, 1, K), bias=b, act=1, out=outbuf)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    row = {"shape": name, "M": M, "N": N, "K": K, "us": round(us, 1), "tflops_fp32_equiv": round(2.0 * M * N * K / us / 1e6, 1),
           "max_abs_err": float(err.max()), "max_err_over_max_ref": float(err.max() / ref.abs().max()),
           "rms_err_over_rms_ref": float((err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))}
    out["shapes"].append(row)
    print(json.dumps(row), flush=True)
    keep.append(w)
    del x, y, ref, err, outbuf
json.dump(out, open(os.path.join("gpurun_out", f"split_gemm_{mode}.json"), "w"), indent=1)
