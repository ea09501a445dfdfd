"""Dev tool: an 8-frame window (160x160) cut into P-frame passes on L lanes: frames/s for (P, L) in 8x1, 4x2, 2x4, 1x4 -- does running
two half-size passes side by side beat one full-size pass?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth, train_utils
from densefusion_amd.native_train import Lanes, NativeTrainer

K, N, M, acc = 21, 1000, 500, int(os.environ.get("ACC", "8"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
objs = []
for j in range(acc):
    o = synth.make_object(500 + j, 160, 160, N, K, M)
    o["obj"][0] = [12, 3, 15, 7][j % 4]
    objs.append(o)
sym = [int(o["obj"][0]) in (12, 15, 18, 19, 20) for o in objs]
fr = {k: torch.stack([torch.from_numpy(o[k]) for o in objs]).to(dev) for k in ("img", "cloud", "choose", "obj", "target", "model_points")}
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()}
for P, n in ((acc, 1), (acc // 2, 2), (acc // 4, 4), (acc // 2, 1)):
    tr = NativeTrainer("posenet", N, K, dev)
    tr.load_state_dict(sd)
    opt = train_utils.FlatAdam(tr, lr=1e-4)
    lanes = Lanes(tr, n)

    def window():
        jobs = [(lambda lane, i=i: lane.step_posenet(*[fr[k][i:i + P] for k in ("img", "cloud", "choose", "obj", "target", "model_points")], sym[i:i + P], 0.015,
                                                     dropout=True)) for i in range(0, acc, P)]
        lanes.run(jobs)
        opt.step(grad_scale=1.0 / acc); tr.zero_grad()

    window(); window(); torch.cuda.synchronize()
    rates = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(6):
            window()
        torch.cuda.synchronize()
        rates.append(round(6 * acc / (time.perf_counter() - t0), 1))
    print(f"{P} frames per pass on {n} lane(s): {rates} frames/s", flush=True)
    lanes.close()
    del lanes, tr, opt
