"""Dev tool: where a refiner-phase frame's time goes (frozen estimator forward, Loss with refine=True, native refiner step) at 1 and at B frames per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth
from densefusion_amd.lib.network import PoseNet
from densefusion_amd.lib.loss import Loss
from densefusion_amd.native_train import NativeTrainer

K, N, M = 21, 1000, 500
dev = torch.device("cuda")
est = PoseNet(N, K); est.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()}); est = est.to(dev).eval()
tr = NativeTrainer("refiner", N, K, dev)
tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K), 14).items()})
crit = Loss(M, [12, 15])

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for B in (1, 4, 16):
    objs = [synth.make_object(500 + j, 160, 160, N, K, M) for j in range(B)]
    fr = {k: torch.stack([torch.from_numpy(o[k]) for o in objs]).to(dev) for k in ("img", "cloud", "choose", "obj", "target", "model_points")}
    with torch.no_grad():
        t_est = timeit(lambda: est(fr["img"], fr["cloud"], fr["choose"], fr["obj"]))
        pr, pt, pc, emb = est(fr["img"], fr["cloud"], fr["choose"], fr["obj"])
        def losses():
            out = []
            for b in range(B):
                out.append(crit(pr[b:b+1], pt[b:b+1], pc[b:b+1], fr["target"][b:b+1], fr["model_points"][b:b+1], fr["obj"][b:b+1], fr["cloud"][b:b+1], 0.015, True))
            return out
        t_loss = timeit(losses)
        o = losses()
        npts, ntg = torch.cat([x[2] for x in o]), torch.cat([x[3] for x in o])
    sym = [False] * B
    t_ref = timeit(lambda: tr.step_refiner(npts, emb, fr["obj"], ntg, fr["model_points"], sym))
    print(f"B={B}: estimator {t_est:.3f} ms, loss(refine) x{B} {t_loss:.3f} ms, refiner step {t_ref:.3f} ms -> per frame with 2 iterations {(t_est + t_loss + 2 * t_ref) / B:.3f} ms", flush=True)
