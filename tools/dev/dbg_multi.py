import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from densefusion_amd import synth
from densefusion_amd.native_train import NativeTrainer
DEV='cuda:0'
K, N, M = 3, 128, 60
sizes = [(40, 80), (160, 160), (80, 80), (40, 80), (120, 160)]
sd = synth.make_state_dict(synth.posenet_spec(K), 23)
objs = [synth.make_object(900 + i, h, w, N, K, num_points_mesh=M) for i, (h, w) in enumerate(sizes)]
for i, o in enumerate(objs): o["obj"][0] = i % K
frames = [dict(img=torch.from_numpy(o["img"]).to(DEV), cloud=torch.from_numpy(o["cloud"]).to(DEV), choose=torch.from_numpy(o["choose"]).to(DEV),
               obj=torch.from_numpy(o["obj"]).to(DEV), target=torch.from_numpy(o["target"]).to(DEV), model_points=torch.from_numpy(o["model_points"]).to(DEV),
               symmetric=int(o["obj"][0]) == 1) for o in objs]
tr = NativeTrainer("posenet", N, K, DEV)
tr.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
out, order = tr.step_posenet_window(frames, 0.015, dropout=False)
gm = tr.grad_dict(); tr.zero_grad()
for row, j in enumerate(order):
    f = frames[j]
    tr.step_posenet(f["img"][None], f["cloud"][None], f["choose"].reshape(1, -1), f["obj"].reshape(1), f["target"][None], f["model_points"][None], [f["symmetric"]], 0.015, dropout=False)
gs = tr.grad_dict()
devs = []
for k, v in gs.items():
    sc = max(float(v.abs().max()), 1e-12)
    devs.append((float((gm[k]-v).abs().max())/sc, k))
for d, k in sorted(devs, reverse=True)[:14]: print(f"{d:.2e} {k}")
import math
num = sum(float(((gm[k]-v).double()**2).sum()) for k, v in gs.items()); den = sum(float((v.double()**2).sum()) for k, v in gs.items())
print("relative L2 of the whole gradient:", math.sqrt(num/den))
per = sorted((math.sqrt(float(((gm[k]-v).double()**2).sum())/max(float((v.double()**2).sum()),1e-30)), k) for k, v in gs.items() if float(v.abs().max())>0)
print("worst per-tensor relative L2:", per[-3:])
