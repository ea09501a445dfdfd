"""Dev tool: timings of the widened rows (f1 input preparation, f2 YCB distances, a12-a13 losses fwd/bwd, f4 training
step) on the GPU next to their CPU oracles on the host -- the numbers quoted in DESIGN.md section 6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from densefusion_amd import synth
from densefusion_amd.lib import preprocess as pp, ycb_eval
from densefusion_amd.lib.loss import Loss
from oracle import loss_ref, preprocess_ref, ycb_metric


def gpu_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def cpu_time(fn, n=3):
    fn(); t = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t) / n * 1e3


def main():
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    rng = np.random.default_rng(0)
    # f1: 10 objects of 160x160 from 2 frames
    F, B, N = 2, 10, 1000
    rgb = rng.integers(0, 256, (F, 480, 640, 3), dtype=np.uint8)
    depth = rng.integers(3000, 15000, (F, 480, 640)).astype(np.uint16)
    label = np.zeros((F, 480, 640), dtype=np.int32)
    objs = []
    for i in range(B):
        f, r0, c0 = i % F, 20 + 30 * (i // 2), 40 + 40 * (i // 2)
        label[f, r0:r0 + 160, c0:c0 + 160][rng.random((160, 160)) < 0.6] = i + 1
        objs.append((f, i + 1, (r0, r0 + 160, c0, c0 + 160), 7 + i))
    d = [torch.from_numpy(a).cuda() for a in (rgb, depth.view(np.int16), label)]
    t_g = gpu_time(lambda: pp.preprocess_objects(d[0], d[1], d[2], objs, N))
    t_c = cpu_time(lambda: [preprocess_ref.prepare_object(rgb[f], depth[f], label[f], it, bb, N, sd, pp.YCB_CAM) for f, it, bb, sd in objs])
    print(f"f1 input preparation, 10 objects 160x160: GPU {t_g*1e3:.0f} us   CPU oracle {t_c:.1f} ms   x{t_c/t_g:.0f}")
    # f2: 100 objects x 2620 model points
    B2, M2 = 100, 2620
    pts = (rng.random((B2, M2, 3)) - 0.5) * 0.2
    est = np.stack([ycb_eval.pose_to_rt(np.r_[synth.random_unit_quaternion(rng), rng.standard_normal(3) * 0.1]) for _ in range(B2)])
    gt = np.stack([ycb_eval.pose_to_rt(np.r_[synth.random_unit_quaternion(rng), rng.standard_normal(3) * 0.1]) for _ in range(B2)])
    dd = [torch.from_numpy(a).cuda() for a in (est, gt, pts)]
    t_g = gpu_time(lambda: ycb_eval.ycb_distances(*dd), 5)
    t_c = cpu_time(lambda: [ycb_metric.adi(est[b], gt[b], pts[b].T) for b in range(10)], 1) * 10
    print(f"f2 YCB add/adi, 100 objects x 2620 pts (fp64): GPU {t_g:.2f} ms   CPU oracle (numpy brute force) {t_c:.0f} ms   x{t_c/t_g:.0f}")
    # a12: symmetric loss forward + backward, N=1000, M=500 (YCB training size: 250M pairs)
    Nn, M = 1000, 500
    o = synth.make_object(1, 160, 160, Nn, 21, M)
    q = rng.standard_normal((1, Nn, 4)).astype(np.float32); ptt = (rng.standard_normal((1, Nn, 3)) * 0.03).astype(np.float32)
    pc = rng.uniform(0.05, 0.95, (1, Nn, 1)).astype(np.float32)
    C = torch.from_numpy
    idx = torch.tensor([[12]])
    g = [C(a).cuda() for a in (q, ptt, pc, o["target"][None], o["model_points"][None], o["cloud"][None])]
    crit = Loss(M, [12])
    def gl():
        a, b, c = g[0].clone().requires_grad_(), g[1].clone().requires_grad_(), g[2].clone().requires_grad_()
        crit(a, b, c, g[3], g[4], idx.cuda(), g[5], 0.015, False)[0].backward()
    t_g = gpu_time(gl, 10)
    def cl():
        a, b, c = C(q).clone().requires_grad_(), C(ptt).clone().requires_grad_(), C(pc).clone().requires_grad_()
        loss_ref.loss_calculation(a, b, c, C(o["target"][None]), C(o["model_points"][None]), idx, C(o["cloud"][None]), 0.015, False, M, [12])[0].backward()
    t_c = cpu_time(cl, 2)
    print(f"a12 symmetric loss fwd+bwd, N=1000 M=500: GPU {t_g*1e3:.0f} us   CPU oracle {t_c:.0f} ms   x{t_c/t_g:.0f}")


if __name__ == "__main__":
    main()
