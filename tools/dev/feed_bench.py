"""Dev tool: does the trainer's data feed keep up?  Frames/s of the native PoseNet training loop on a fabricated LineMOD tree (PNG
decoding, gt.yml, .ply models: densefusion_amd.datasets.linemod) fed through train_utils.Prefetcher with 0 / 4 / 8 worker threads
and with 8 / 12 worker processes, against the same loop over frames that already sit in device memory.
usage: feed_bench.py TREE_ROOT"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densefusion_amd import synth, train_utils
from densefusion_amd.native_train import Lanes, NativeTrainer


def run(root, workers_list=(0, 4, 8), frames=96, lanes_n=4, out=print, processes_list=(8, 12)):
    from densefusion_amd.datasets.linemod.dataset import PoseDataset
    dev = torch.device("cuda")
    ds = PoseDataset("train", 500, False, root, 0.0, False)
    K, N = 13, 500
    tr = NativeTrainer("posenet", N, K, dev)
    tr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 21).items()})
    opt = train_utils.FlatAdam(tr, lr=1e-4)
    lanes = Lanes(tr, lanes_n)
    sym = ds.get_sym_list()
    order = [i % len(ds) for i in range(frames)]

    def loop(items):
        window, n = [], 0
        for it in items:
            if it[0].dim() == 1:
                continue
            window.append(it)
            if len(window) == 8:
                jobs = [(lambda lane, f=f: lane.step_posenet(f[2][None], f[0][None], f[1], f[5], f[3][None], f[4][None],
                                                              [train_utils.host_index(f[5]) in sym], 0.015)) for f in window]
                lanes.run(jobs)
                opt.step(grad_scale=1.0 / 8); tr.zero_grad()
                n += len(window)
                window = []
        torch.cuda.synchronize()
        return n

    resident = [ds[i] for i in order]
    loop(resident[:16])                        # warm-up: workspaces, weight copies
    t0 = time.perf_counter(); n = loop(resident); base = n / (time.perf_counter() - t0)
    res = {"resident_frames_per_s": round(base, 1)}
    for w in workers_list:
        t0 = time.perf_counter()
        n = loop(train_utils.Prefetcher(ds, order, dev, workers=w))
        res[f"workers_{w}_frames_per_s"] = round(n / (time.perf_counter() - t0), 1)
    for w in processes_list:
        pf = train_utils.Prefetcher(ds, order, dev, workers=0, processes=w)
        loop(pf.set_order(order[:16]))           # the worker processes start (imports) outside the timed pass, as in a long run
        t0 = time.perf_counter()
        n = loop(pf.set_order(order))
        res[f"processes_{w}_frames_per_s"] = round(n / (time.perf_counter() - t0), 1)
        pf.close()
    lanes.close()
    out(res)
    return res


if __name__ == "__main__":
    run(sys.argv[1])
