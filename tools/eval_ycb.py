#!/usr/bin/env python
"""YCB-Video evaluation driver -- same job and flags as the reference's tools/eval_ycb.py, on the HIP path.

    python tools/eval_ycb.py --dataset_root <YCB_Video_Dataset> --model <pose_model.pth> --refine_model <refine.pth>

For every keyframe of ``test_data_list.txt`` and every PoseCNN detection: snap the ROI (get_bbox), prepare
the inputs ON THE DEVICE (mask, choose, cloud, normalised crop -- densefusion_amd.lib.preprocess), run
PoseNet + arg-max selection + ``iteration`` refine steps as one device call (PoseEstimator), and write
``{'poses': [n,7]}`` to ``Densefusion_wo_refine_result/%04d.mat`` and ``Densefusion_iterative_result/%04d.mat``
(tools/eval_ycb.py:136-240), which the YCB toolbox scripts -- or densefusion_amd.lib.ycb_eval -- consume.

Differences from the reference script: constants are flags with the reference's values as defaults; the
random pixel subset follows the documented key rule instead of np.random (include/dfusion.h); objects of
one frame that share a crop size go through the network as one batch; a detection without mask pixels is
reported as lost (zero pose), like the reference's ZeroDivisionError branch (:234-237).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import scipy.io as scio
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densefusion_amd.lib import preprocess as pp  # noqa: E402
from densefusion_amd.lib.network import PoseEstimator, PoseNet, PoseRefineNet  # noqa: E402


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
    pe = PoseEstimator(estimator, refiner)

    testlist = read_lines(os.path.join(opt.dataset_config_dir, "test_data_list.txt"))
    if opt.max_frames > 0:
        testlist = testlist[:opt.max_frames]
    print(len(testlist))
    os.makedirs(opt.result_wo_refine_dir, exist_ok=True)
    os.makedirs(opt.result_refine_dir, exist_ok=True)

    for now, rel in enumerate(testlist):
        if now % world != rank:
            continue
        rgb = np.array(Image.open("{0}/{1}-color.png".format(opt.dataset_root, rel)))[:, :, :3]
        depth = np.array(Image.open("{0}/{1}-depth.png".format(opt.dataset_root, rel))).astype(np.uint16)
        meta = scio.loadmat("{0}/results_PoseCNN_RSS2018/{1}.mat".format(opt.ycb_toolbox_dir, "%06d" % now))
        label = np.array(meta["labels"]).astype(np.int32)
        rois = np.array(meta["rois"])
        d_rgb = torch.from_numpy(np.ascontiguousarray(rgb))[None].to(dev)
        d_depth = torch.from_numpy(depth.view(np.int16))[None].to(dev)
        d_label = torch.from_numpy(label)[None].to(dev)
        n = rois.shape[0]
        wo = np.zeros((n, 7))
        refined = np.zeros((n, 7))
        groups = {}
        for idx in range(n):
            bb = pp.get_bbox(rois[idx])
            groups.setdefault((bb[1] - bb[0], bb[3] - bb[2]), []).append((idx, int(rois[idx][1]), bb))
        for (H, W), members in groups.items():
            objs = [(0, itemid, bb, opt.seed + now * 64 + idx) for idx, itemid, bb in members]
            img, cloud, choose, count = pp.preprocess_objects(d_rgb, d_depth, d_label, objs, opt.num_points)
            index = torch.tensor([itemid - 1 for _, itemid, _ in members], dtype=torch.int64, device=dev)
            p_wo, p_ref = pe.estimate(img, cloud, choose, index, opt.iteration)
            p_wo, p_ref, count = p_wo.cpu().numpy(), p_ref.cpu().numpy(), count.cpu().numpy()
            for k, (idx, itemid, _) in enumerate(members):
                if count[k] == 0:
                    print("PoseCNN Detector Lost {0} at No.{1} keyframe".format(itemid, now))
                    continue                                  # zero pose rows, like the reference
                wo[idx], refined[idx] = p_wo[k], p_ref[k]
        scio.savemat("{0}/{1}.mat".format(opt.result_wo_refine_dir, "%04d" % now), {"poses": wo.tolist()})
        scio.savemat("{0}/{1}.mat".format(opt.result_refine_dir, "%04d" % now), {"poses": refined.tolist()})
        print("Finish No.{0} keyframe".format(now))


if __name__ == "__main__":
    main()
