"""Dev tool: training-step throughput (frames/s) on synthetic YCB-shaped frames (BASELINE configs[3] per GPU:
K=21, N=1000, M=500, 8 frames accumulated per optimizer step, 5/21 objects symmetric).
usage: train_bench.py [frames_per_pass]   -- 1 (default) = the reference's bs = 1 passes; P > 1 = P same-size frames per
differentiable pass (same gradients, larger GEMMs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd import synth, train_ops, train_utils
from densefusion_amd.lib.loss import Loss
from densefusion_amd.lib.network import PoseNet

def main():
    K, N, M, acc = 21, 1000, 500, 8
    dev = torch.device("cuda")
    net = PoseNet(N, K)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
    net.to(dev).train()
    flat = train_utils.FlatParams(net); opt = train_utils.FlatAdam(flat, lr=1e-4)
    crit = Loss(M, [12, 15, 18, 19, 20])
    crops = [(80, 80), (120, 120), (120, 160), (160, 160), (160, 200), (200, 240), (240, 320)]
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    groups = []                                               # each group: P frames of one crop size
    for gi in range(16 // P if P > 1 else 16):
        H, W = crops[gi % len(crops)]
        fr = []
        for j in range(P):
            o = synth.make_object(500 + gi * P + j, H, W, N, K, M)
            o["obj"][0] = [12, 3, 15, 7][(gi + j) % 4]          # half of the frames symmetric (KNN loss branch)
            fd = {k: torch.from_numpy(v).to(dev) for k, v in o.items()}
            train_utils.with_host_index(fd["obj"], o["obj"])
            fr.append(fd)
        groups.append(fr)
    def step(fr):
        with train_ops.splitk_scope(dev):
            return _step(fr)

    def _step(fr):
        img = torch.stack([f["img"] for f in fr]); cloud = torch.stack([f["cloud"] for f in fr])
        choose = torch.stack([f["choose"] for f in fr]); obj = torch.stack([f["obj"] for f in fr])
        r, t, c, emb = net(img, cloud, choose, obj)
        loss = 0
        for b, f in enumerate(fr):
            loss = loss + crit(r[b:b + 1], t[b:b + 1], c[b:b + 1], f["target"][None], f["model_points"][None], f["obj"], f["cloud"][None],
                               0.015, False)[0]
        loss.backward()
        return loss
    for fr in groups[:2]: step(fr)
    opt.step(); flat.zero_grad(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for rep in range(2):
        for fr in groups:
            step(fr); n += len(fr)
            if n % acc == 0:
                train_utils.allreduce_gradients(flat); opt.step(); flat.zero_grad()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"training: {n/dt:.1f} frames/s ({dt/n*1e3:.1f} ms per frame fwd+bwd, {P} frame(s) per pass, optimizer step every {acc} frames)")

if __name__ == "__main__":
    main()
