#!/usr/bin/env python
"""LineMOD evaluation driver -- the job of the reference's tools/eval_linemod.py on the HIP path.

    python tools/eval_linemod.py --dataset_root <Linemod_preprocessed> --model <pose_model.pth> --refine_model <refine.pth>

For every test frame: PoseNet -> arg-max pose -> ``iteration`` (4) refine steps in one device call
(PoseEstimator), then ADD (ADD-S through the fused 1-NN for the symmetric objects eggbox / glue) on the
device (``add_metric``) against ``0.1 x diameter`` from ``models_info.yml`` (tools/eval_linemod.py:57-61,
118-139); per-object and overall success rates go to ``eval_result_logs.txt`` in the reference's format.

Frames come from ``densefusion_amd.datasets.linemod.dataset.PoseDataset`` (host PNG / yml / ply decoding, mask ->
choose -> cloud -> crop on the device); any object with the same ``__getitem__`` 6-tuple, ``get_sym_list()`` and
``get_num_points_mesh()`` can be injected instead (``main(testdataset=...)``).
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densefusion_amd.lib.metric import add_metric  # noqa: E402
from densefusion_amd.lib.network import PoseEstimator, PoseNet, PoseRefineNet  # noqa: E402

OBJLIST = [1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15]


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset_root", type=str, default="", help="dataset root dir")
    ap.add_argument("--model", type=str, default="", help="resume PoseNet model")
    ap.add_argument("--refine_model", type=str, default="", help="resume PoseRefineNet model")
    ap.add_argument("--dataset_config_dir", type=str, default="datasets/linemod/dataset_config")
    ap.add_argument("--output_result_dir", type=str, default="experiments/eval_result/linemod")
    ap.add_argument("--num_points", type=int, default=500)
    ap.add_argument("--iteration", type=int, default=4)
    ap.add_argument("--max_frames", type=int, default=0)
    ap.add_argument("--window", type=int, default=64, help="test frames per device call (1 = frame by frame, like the reference)")
    ap.add_argument("--workers", type=int, default=8, help="frames are fetched ahead of the device calls by this many workers (0 = fetch in the loop)")
    ap.add_argument("--feed", type=str, default="threads", choices=["processes", "threads"],
                    help="threads: worker threads of this process fetch (decode, upload, prepare) ahead of the device calls: 112 -> 304 frames/s "
                         "on a fabricated tree; processes: PNG decoding and box finding in worker processes (the loader's host_item) -- their "
                         "start-up (an interpreter + torch import each) only pays off on long runs")
    return ap


def evaluate(testdataset, estimator, refiner, diameter, opt, fw=None):
    """The loop of tools/eval_linemod.py:68-139; returns (success_count, num_count) per object.  Frames are taken ``--window`` at
    a time: the crops of a window are bucketed by size and run through PoseNet -> arg-max pose -> refine loop as ONE device
    call (``estimate_multi``), the ADD / ADD-S distances of the window as one ``add_metric`` launch; the log lines come out in
    frame order and do not depend on the window (per-object results are bit-identical to frame-by-frame calls)."""
    num_objects = len(diameter)
    pe = PoseEstimator(estimator, refiner)
    sym_list = testdataset.get_sym_list()
    success_count, num_count = [0] * num_objects, [0] * num_objects
    say = (lambda m: (print(m), fw.write(m + "\n"))) if fw else print
    n = len(testdataset) if opt.max_frames <= 0 else min(opt.max_frames, len(testdataset))
    dev = torch.device("cuda")
    window = max(1, getattr(opt, "window", 1))
    workers = max(0, getattr(opt, "workers", 0))
    from densefusion_amd.train_utils import Prefetcher
    feed = Prefetcher(testdataset, range(n), dev, workers=workers, processes=workers if getattr(opt, "feed", "threads") == "processes" else 0)
    frames = iter(feed)
    for w0 in range(0, n, window):
        items = [(i, next(frames)) for i in range(w0, min(n, w0 + window))]
        live = [(i, it) for i, it in items if it[0].dim() != 1]       # others: the loader's "no mask pixel" sentinel (dataset.py:135-137)
        dist = {}
        if live:
            buckets = {}
            for k, (i, it) in enumerate(live):
                buckets.setdefault(tuple(it[2].shape[-2:]), []).append(k)
            order = [k for _, ks in sorted(buckets.items()) for k in ks]
            imgs = [torch.stack([live[k][1][2] for k in ks]).to(dev) for _, ks in sorted(buckets.items())]
            cat = lambda j: torch.stack([live[k][1][j] for k in order]).to(dev)
            points, target, model_points = cat(0), cat(3), cat(4)
            choose = torch.stack([live[k][1][1].reshape(-1) for k in order]).to(dev)
            idx = torch.stack([live[k][1][5].reshape(-1)[0] for k in order]).to(dev)
            _, pose = pe.estimate_multi(imgs, points, choose, idx, opt.iteration)
            objs = [int(live[k][1][5].reshape(-1)[0]) for k in order]
            dis = add_metric(pose, model_points, target, [1 if o in sym_list else 0 for o in objs]).cpu().tolist()
            dist = {live[k][0]: (objs[j], dis[j]) for j, k in enumerate(order)}
        for i, it in items:
            if i not in dist:
                say("No.{0} NOT Pass! Lost detection!".format(i))
                continue
            obj, d = dist[i]
            if d < diameter[obj]:
                success_count[obj] += 1
                say("No.{0} Pass! Distance: {1}".format(i, d))
            else:
                say("No.{0} NOT Pass! Distance: {1}".format(i, d))
            num_count[obj] += 1
    feed.close()
    return success_count, num_count


def main(argv=None, testdataset=None):
    opt = build_parser().parse_args(argv)
    num_objects = len(OBJLIST)
    estimator = PoseNet(num_points=opt.num_points, num_obj=num_objects).cuda()
    refiner = PoseRefineNet(num_points=opt.num_points, num_obj=num_objects).cuda()
    estimator.load_state_dict(torch.load(opt.model, map_location="cuda", weights_only=True))
    refiner.load_state_dict(torch.load(opt.refine_model, map_location="cuda", weights_only=True))
    estimator.eval(); refiner.eval()
    if testdataset is None:
        from densefusion_amd.datasets.linemod.dataset import PoseDataset as PoseDataset_linemod
        testdataset = PoseDataset_linemod("eval", opt.num_points, False, opt.dataset_root, 0.0, True)
    with open("{0}/models_info.yml".format(opt.dataset_config_dir), "r") as f:
        meta = yaml.safe_load(f)
    diameter = [meta[obj]["diameter"] / 1000.0 * 0.1 for obj in OBJLIST]
    print(diameter)
    os.makedirs(opt.output_result_dir, exist_ok=True)
    with open("{0}/eval_result_logs.txt".format(opt.output_result_dir), "w") as fw:
        success_count, num_count = evaluate(testdataset, estimator, refiner, diameter, opt, fw)
        for i in range(num_objects):
            if num_count[i]:
                m = "Object {0} success rate: {1}".format(OBJLIST[i], float(success_count[i]) / num_count[i])
                print(m); fw.write(m + "\n")
        m = "ALL success rate: {0}".format(float(sum(success_count)) / max(1, sum(num_count)))
        print(m); fw.write(m + "\n")
    return success_count, num_count


if __name__ == "__main__":
    main()
