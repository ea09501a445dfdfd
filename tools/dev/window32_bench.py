"""Dev tool: frames/s of the native step on a 32-frame window of mixed crop sizes (this fork's default batch), for A/B of development switches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth, train_utils
from densefusion_amd.native_train import NativeTrainer
K, N, M = 21, 1000, 500
dev = torch.device("cuda")
CROPS = [(80, 80), (120, 120), (120, 160), (160, 160), (160, 200), (200, 240), (240, 320)]
frames = []
for j in range(32):
    H, W = CROPS[j % 7]
    o = synth.make_object(800 + j, H, W, N, K, M)
    o["obj"][0] = [12, 3, 15, 7][j % 4]
    frames.append(dict(img=torch.from_numpy(o["img"]).to(dev), cloud=torch.from_numpy(o["cloud"]).to(dev), choose=torch.from_numpy(o["choose"]).to(dev),
                       obj=torch.from_numpy(o["obj"]).to(dev), target=torch.from_numpy(o["target"]).to(dev), model_points=torch.from_numpy(o["model_points"]).to(dev),
                       symmetric=int(o["obj"][0]) in (12, 15)))
tr = NativeTrainer("posenet", N, K, dev)
tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
opt = train_utils.FlatAdam(tr, lr=1e-4)
def win():
    tr.step_posenet_window(frames, 0.015); opt.step(grad_scale=1 / 32); tr.zero_grad()
win(); win(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): win()
torch.cuda.synchronize()
print(f"window of 32: {5 * 32 / (time.perf_counter() - t0):.1f} frames/s")
