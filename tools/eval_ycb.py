#!/usr/bin/env python
"""YCB-Video evaluation driver -- same job and flags as the reference's tools/eval_ycb.py, on the HIP path.

    python tools/eval_ycb.py --dataset_root <YCB_Video_Dataset> --model <pose_model.pth> --refine_model <refine.pth>

For every keyframe of ``test_data_list.txt`` and every PoseCNN detection: snap the ROI (get_bbox), prepare
the inputs ON THE DEVICE (mask, choose, cloud, normalised crop -- densefusion_amd.lib.preprocess), run
PoseNet + arg-max selection + ``iteration`` refine steps as one device call (PoseEstimator), and write
``{'poses': [n,7]}`` to ``Densefusion_wo_refine_result/%04d.mat`` and ``Densefusion_iterative_result/%04d.mat``
(tools/eval_ycb.py:136-240), which the YCB toolbox scripts -- or densefusion_amd.lib.ycb_eval -- consume.

Differences from the reference script: constants are flags with the reference's values as defaults; the
random pixel subset follows the documented key rule instead of np.random (include/dfusion.h); a detection without
mask pixels, or with a degenerate box, is reported as lost (zero pose), like the reference's ZeroDivisionError branch
(:234-237).  Throughput: the keyframes are taken ``--window`` at a time (densefusion_amd.lib.eval_window): PNG decoding
runs in ``--workers`` threads one window ahead, a window's frames go up in one copy that overlaps the previous window's
compute, and ALL detections of the window -- bucketed by snapped crop size across its frames -- run through the network
in one device call.  The result files do not depend on the window length (tests/test_eval_ycb_tool_gpu.py).
"""
from __future__ import annotations

import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.io as scio
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densefusion_amd.lib.eval_window import WindowEstimator  # noqa: E402
from densefusion_amd.lib.network import PoseNet, PoseRefineNet  # noqa: E402


def read_lines(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip()]


def load_points_xyz(path):
    """models/<class>/points.xyz: one 'x y z' per line (eval_ycb.py:123-133)."""
    return np.loadtxt(path, dtype=np.float64).reshape(-1, 3)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset_root", type=str, default="", help="dataset root dir")
    ap.add_argument("--model", type=str, default="", help="resume PoseNet model")
    ap.add_argument("--refine_model", type=str, default="", help="resume PoseRefineNet model")
    ap.add_argument("--dataset_config_dir", type=str, default="datasets/ycb/dataset_config")
    ap.add_argument("--ycb_toolbox_dir", type=str, default="YCB_Video_toolbox")
    ap.add_argument("--result_wo_refine_dir", type=str, default="experiments/eval_result/ycb/Densefusion_wo_refine_result")
    ap.add_argument("--result_refine_dir", type=str, default="experiments/eval_result/ycb/Densefusion_iterative_result")
    ap.add_argument("--num_obj", type=int, default=21)
    ap.add_argument("--num_points", type=int, default=1000)
    ap.add_argument("--iteration", type=int, default=2)
    ap.add_argument("--max_frames", type=int, default=0, help="0 = all keyframes of test_data_list.txt")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--window", type=int, default=32, help="keyframes per device call (1 = frame by frame)")
    ap.add_argument("--workers", type=int, default=8, help="PNG / .mat reader threads")
    ap.add_argument("--depth", type=int, default=4, help="windows in flight on the device (own stream and workspace each)")
    return ap


def main(argv=None):
    opt = build_parser().parse_args(argv)
    # frames shard over the GPUs of a node without any collective: under `python -m torch.distributed.run --nproc-per-node N`
    # rank r takes the keyframes r, r + N, ... and writes their result files (every frame is independent)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    estimator = PoseNet(num_points=opt.num_points, num_obj=opt.num_obj)
    estimator.to(dev)
    estimator.load_state_dict(torch.load(opt.model, map_location=dev, weights_only=True))
    estimator.eval()
    refiner = PoseRefineNet(num_points=opt.num_points, num_obj=opt.num_obj)
    refiner.to(dev)
    refiner.load_state_dict(torch.load(opt.refine_model, map_location=dev, weights_only=True))
    refiner.eval()

    testlist = read_lines(os.path.join(opt.dataset_config_dir, "test_data_list.txt"))
    if opt.max_frames > 0:
        testlist = testlist[:opt.max_frames]
    print(len(testlist))
    os.makedirs(opt.result_wo_refine_dir, exist_ok=True)
    os.makedirs(opt.result_refine_dir, exist_ok=True)
    mine = [now for now in range(len(testlist)) if now % world == rank]
    window = max(1, opt.window)
    windows = [mine[i:i + window] for i in range(0, len(mine), window)]
    if not windows:
        return
    IH, IW = 480, 640
    depth = max(1, opt.depth)
    we = WindowEstimator(estimator, refiner, opt.num_points, opt.iteration, window, (IH, IW), depth=depth)
    # pinned host slots: `depth` windows whose uploads may still be in flight + the one the reader threads are filling
    NH = depth + 2
    host = [dict(rgb=torch.empty(window, IH, IW, 3, dtype=torch.uint8).pin_memory(),
                 depth=torch.empty(window, IH, IW, dtype=torch.int16).pin_memory(),
                 label=torch.empty(window, IH, IW, dtype=torch.int32).pin_memory()) for _ in range(NH)]
    pool = ThreadPoolExecutor(max_workers=max(1, opt.workers))

    def read_frame(slot, f, now):
        rel = testlist[now]
        h = host[slot]
        np.copyto(h["rgb"][f].numpy(), np.array(Image.open("{0}/{1}-color.png".format(opt.dataset_root, rel)))[:, :, :3])
        np.copyto(h["depth"][f].numpy(), np.array(Image.open("{0}/{1}-depth.png".format(opt.dataset_root, rel))).astype(np.uint16).view(np.int16))
        meta = scio.loadmat("{0}/results_PoseCNN_RSS2018/{1}.mat".format(opt.ycb_toolbox_dir, "%06d" % now))
        np.copyto(h["label"][f].numpy(), np.array(meta["labels"]).astype(np.int32))
        return np.array(meta["rois"])

    def start_read(wi):
        return [pool.submit(read_frame, wi % NH, f, now) for f, now in enumerate(windows[wi])]

    def finish(wi, handle, rois_per_frame):
        wo, refined, lost = we.collect(handle)
        k = 0
        for now, rois in zip(windows[wi], rois_per_frame):
            n = rois.shape[0]
            for idx in range(n):
                if lost[k + idx]:
                    print("PoseCNN Detector Lost {0} at No.{1} keyframe".format(int(rois[idx][1]), now))
            scio.savemat("{0}/{1}.mat".format(opt.result_wo_refine_dir, "%04d" % now), {"poses": wo[k:k + n].tolist()})
            scio.savemat("{0}/{1}.mat".format(opt.result_refine_dir, "%04d" % now), {"poses": refined[k:k + n].tolist()})
            print("Finish No.{0} keyframe".format(now))
            k += n

    from collections import deque
    reads = {0: start_read(0)}
    pending = deque()
    for wi in range(len(windows)):
        rois_per_frame = [f.result() for f in reads.pop(wi)]
        while len(pending) >= depth:
            finish(*pending.popleft())                   # oldest window's results (frees its host slot), others keep running
        if wi + 1 < len(windows):
            reads[wi + 1] = start_read(wi + 1)            # decoded while this window uploads and computes
        F = len(windows[wi])
        h = host[wi % NH]
        dets = [(f, int(rois[idx][1]), rois[idx], opt.seed + now * 64 + idx)
                for f, (now, rois) in enumerate(zip(windows[wi], rois_per_frame)) for idx in range(rois.shape[0])]
        handle = we.submit(h["rgb"][:F], h["depth"][:F], h["label"][:F], dets)       # enqueued, no host sync
        pending.append((wi, handle, rois_per_frame))
    while pending:
        finish(*pending.popleft())
    pool.shutdown()


if __name__ == "__main__":
    main()
