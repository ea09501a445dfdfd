"""Dev tool (GPU box, development library): every MFMA launch of one native training window with its shape, duration and TFLOP/s, slowest first.
usage: DF_DEV_LIB=1 DF_PROFILE_VERBOSE=1 python tools/dev/train_gemm_list.py [8|mixed|32] 2> gpurun_out/train_gemm_raw.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth
from densefusion_amd.native_train import NativeTrainer

mode = sys.argv[1] if len(sys.argv) > 1 else "8"
K, N, M = 21, 1000, 500
dev = torch.device("cuda")
CROPS = [(80, 80), (120, 120), (120, 160), (160, 160), (160, 200), (200, 240), (240, 320)]
nfr = 32 if mode == "32" else 8
sizes = [(160, 160)] * 8 if mode == "8" else [CROPS[j % len(CROPS)] for j in range(nfr)]
frames = []
for j, (H, W) in enumerate(sizes):
    o = synth.make_object(800 + j, H, W, N, K, M)
    o["obj"][0] = [12, 3, 15, 7][j % 4]
    frames.append(dict(img=torch.from_numpy(o["img"]).to(dev), cloud=torch.from_numpy(o["cloud"]).to(dev), choose=torch.from_numpy(o["choose"]).to(dev),
                       obj=torch.from_numpy(o["obj"]).to(dev), target=torch.from_numpy(o["target"]).to(dev), model_points=torch.from_numpy(o["model_points"]).to(dev),
                       symmetric=int(o["obj"][0]) in (12, 15, 18, 19, 20)))
tr = NativeTrainer("posenet", N, K, dev)
tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
for _ in range(2):
    tr.step_posenet_window(frames, 0.015); tr.zero_grad()
torch.cuda.synchronize()
tr.profile(True)
tr.step_posenet_window(frames, 0.015)
torch.cuda.synchronize()
print(tr.profile_read())
