"""Dev tool: host time of one native training-step call vs its GPU time (B = 1, 160x160), eager and replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth
from densefusion_amd.native_train import NativeTrainer
K, N, M = 21, 1000, 500
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
tr = NativeTrainer("posenet", N, K, dev)
tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
objs = [synth.make_object(500 + j, 160, 160, N, K, M) for j in range(B)]
fr = [torch.stack([torch.from_numpy(o[k]) for o in objs]).to(dev) for k in ("img", "cloud", "choose", "obj", "target", "model_points")]
sym = [j % 2 == 0 for j in range(B)]
for _ in range(3):
    tr.step_posenet(*fr, sym, 0.015)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.step_posenet(*fr, sym, 0.015)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host {1e3*(t1-t0)/n:.2f} ms per call, wall {1e3*(t2-t0)/n:.2f} ms per call (B={B})")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    tr.step_posenet(*fr, sym, 0.015, graph_safe=True, seed=5)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    tr.step_posenet(*fr, sym, 0.015, graph_safe=True, seed=5)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print(f"graph: {1e3*(time.perf_counter()-t0)/n:.2f} ms per replay (includes the per-step weight flips)")
