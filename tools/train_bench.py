"""Dev tool: training-step throughput (frames/s) of the NATIVE step on synthetic YCB-shaped frames (BASELINE configs[3] per GPU:
K=21, N=1000, M=500, 8 frames accumulated per optimizer step, half of the frames symmetric), crop sizes cycled over the bench's seven.
usage: train_bench.py [frames_per_pass] [reps]   -- 1 (default) = the reference's bs = 1 passes; P > 1 = P same-size frames per call.
       train_bench.py mixed [reps]               -- windows of 8 frames of 8 DIFFERENT crop sizes, each window ONE multi-bucket pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densefusion_amd import synth, train_utils
from densefusion_amd.native_train import NativeTrainer

def main():
    K, N, M, acc = 21, 1000, 500, 8
    dev = torch.device("cuda")
    mixed = len(sys.argv) > 1 and sys.argv[1] == "mixed"
    P = 1 if mixed else int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    tr = NativeTrainer("posenet", N, K, dev)
    tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
    opt = train_utils.FlatAdam(tr, lr=1e-4)
    crops = [(160, 160)] if os.environ.get("DF_TB_ONE_SIZE") else [(80, 80), (120, 120), (120, 160), (160, 160), (160, 200), (200, 240), (240, 320)]
    sym_list = [12, 15, 18, 19, 20]
    groups = []
    for gi in range(max(16 // P, len(crops)) if P > 1 else 16):
        H, W = crops[gi % len(crops)]
        objs = []
        for j in range(P):
            o = synth.make_object(500 + gi * P + j, H, W, N, K, M)
            o["obj"][0] = [12, 3, 15, 7][(gi + j) % 4]
            objs.append(o)
        fr = [torch.stack([torch.from_numpy(o[k]) for o in objs]).to(dev) for k in ("img", "cloud", "choose", "obj", "target", "model_points")]
        groups.append((fr, [int(o["obj"][0]) in sym_list for o in objs]))
    if mixed:
        frames = [dict(img=fr[0][0], cloud=fr[1][0], choose=fr[2][0], obj=fr[3][0], target=fr[4][0], model_points=fr[5][0], symmetric=sym[0]) for fr, sym in groups]
        windows = [frames[i:i + acc] for i in range(0, len(frames), acc)]
        for w in windows:
            tr.step_posenet_window(w, 0.015)
            opt.step(grad_scale=1.0 / acc); tr.zero_grad()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for rep in range(reps):
            for w in windows:
                tr.step_posenet_window(w, 0.015); n += len(w)
                opt.step(grad_scale=1.0 / acc); tr.zero_grad()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"training (native step): {n/dt:.1f} frames/s ({dt/n*1e3:.2f} ms per frame fwd+bwd, windows of {acc} mixed-size frames as one multi-bucket pass)")
        return
    for fr, sym in groups:
        tr.step_posenet(*fr, sym, 0.015)
    opt.step(); tr.zero_grad(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for rep in range(reps):
        for fr, sym in groups:
            tr.step_posenet(*fr, sym, 0.015); n += len(sym)
            if n % acc == 0:
                opt.step(grad_scale=1.0 / acc); tr.zero_grad()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"training (native step): {n/dt:.1f} frames/s ({dt/n*1e3:.2f} ms per frame fwd+bwd, {P} frame(s) per pass, optimizer step every {acc} frames)")

if __name__ == "__main__":
    main()
