"""Dev tool: is the 4-lane bs = 1 training rate bound by the host's launches or by the device?  Each lane replays a captured
hipGraph of its one-frame step (no launches from the host) against the eager lanes.  usage: lanes_graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth
from densefusion_amd.native_train import Lanes, NativeTrainer

K, N, M, acc = 21, 1000, 500, 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
objs = [synth.make_object(500 + j, 160, 160, N, K, M) for j in range(acc)]
fr = [{k: torch.from_numpy(o[k])[None].to(dev) for k in ("img", "cloud", "choose", "obj", "target", "model_points")} for o in objs]
tr = NativeTrainer("posenet", N, K, dev)
tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
for n in (4,):
    lanes = Lanes(tr, n)
    args = lambda f: (f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [False], 0.015)
    jobs = [(lambda lane, f=f: lane.step_posenet(*args(f))) for f in fr]
    for _ in range(3):
        lanes.run(jobs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        lanes.run(jobs)
    torch.cuda.synchronize()
    print(f"eager lanes {n}: {6 * acc / (time.perf_counter() - t0):.1f} frames/s")
    # one graph per (lane, frame slot): lane li owns frames li, li + n, ...
    graphs = []
    for li, lane in enumerate(lanes.lanes):
        lane.data, lane.version = tr.data, tr.version
        for j in range(li, acc, n):
            st = lanes.streams[li]
            with torch.cuda.stream(st):
                lane.step_posenet(*args(fr[j]))                  # eager on this stream: flips cached for this version
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                lane.step_posenet(*args(fr[j]))
            graphs.append((li, g))
    torch.cuda.synchronize()

    def window():
        for li, g in graphs:
            with torch.cuda.stream(lanes.streams[li]):
                g.replay()
    for _ in range(3):
        window()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        window()
    torch.cuda.synchronize()
    print(f"graph lanes {n}: {6 * acc / (time.perf_counter() - t0):.1f} frames/s (replays only: no gradient sum / optimizer step)")
    lanes.close()
