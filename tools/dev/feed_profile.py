"""Dev tool: where a LineMOD fetch spends its time (fabricated tree): host decode alone with 1 / 4 / 8 / 16 threads, the device
part alone, the target sampling alone.  usage: feed_profile.py TREE_ROOT"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from concurrent.futures import ThreadPoolExecutor
import torch


def main(root):
    from densefusion_amd.datasets.linemod.dataset import PoseDataset
    ds = PoseDataset("train", 500, False, root, 0.0, False)
    n = min(len(ds), 96)
    idx = [i % len(ds) for i in range(n)]
    ds[0]; torch.cuda.synchronize()
    res = {"cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for w in (1, 4, 8, 16):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(w) as ex:
            list(ex.map(ds._host_frame, idx))
        res[f"host_frame_threads_{w}_fps"] = round(n / (time.perf_counter() - t0), 1)
    host = [ds._host_frame(i) for i in idx]
    t0 = time.perf_counter()
    for h in host:
        ds._targets(h[4], h[5])
    res["targets_fps"] = round(n / (time.perf_counter() - t0), 1)
    t0 = time.perf_counter()
    for i in idx:
        ds[i]
    torch.cuda.synchronize()
    res["getitem_fps"] = round(n / (time.perf_counter() - t0), 1)
    for w in (4, 16):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(w) as ex:
            list(ex.map(ds.__getitem__, idx))
        torch.cuda.synchronize()
        res[f"getitem_threads_{w}_fps"] = round(n / (time.perf_counter() - t0), 1)
    print(res)


if __name__ == "__main__":
    main(sys.argv[1])
