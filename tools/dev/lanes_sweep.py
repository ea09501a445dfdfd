"""Dev tool: frames/s of the native PoseNet training step with the reference's bs = 1 passes on 1 / 2 / 4 / 8 lanes (bench.py's
training workload: 8 frames of 160x160 per optimizer step).  usage: lanes_sweep.py [lane counts ...]   (default 1 2 4 8; the HIP
runtime offers 4 hardware queues: more lanes than that share them)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth, train_utils
from densefusion_amd.native_train import Lanes, NativeTrainer

K, N, M, acc = 21, 1000, 500, 8
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
for n in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    tr = NativeTrainer("posenet", N, K, dev)
    tr.load_state_dict(sd)
    opt = train_utils.FlatAdam(tr, lr=1e-4)
    lanes = Lanes(tr, n)

    def window():
        jobs = [(lambda lane, i=i: lane.step_posenet(*[fr[k][i:i + 1] for k in ("img", "cloud", "choose", "obj", "target", "model_points")], sym[i:i + 1], 0.015,
                                                     dropout=True)) for i in range(acc)]
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
    print(f"lanes {n}: {rates} frames/s", flush=True)
    lanes.close()
    del lanes, tr, opt
