"""A window of keyframes through the hot path in one go -- the throughput form of the loop body of
tools/eval_ycb.py:136-240.

The reference handles one detection at a time (tools/eval_ycb.py:147-237: numpy preparation, 4 uploads, PoseNet, a
device->host sync, ``iteration`` refiner round trips).  ``WindowEstimator`` takes ALL detections of a window of frames:

  * the frames (colour, depth, PoseCNN label map) go up once, from pinned host buffers, on a copy stream -- the upload of
    window i+1 runs while window i computes;
  * detections are bucketed by their snapped box size (``get_bbox``, eval_ycb.py:54-90) across the whole window; each
    bucket's inputs are prepared on the device in one launch (``preprocess_objects``);
  * all buckets go through PoseNet -> pose selection -> refine loop as ONE device call (``df_estimate_poses_multi``):
    every launch that does not depend on the crop size covers the whole window;
  * one device->host copy returns the [n,7] poses of the window;
  * up to ``depth`` windows are in flight, each on its own stream with its own workspace, so one window's memory-bound
    kernels (transforms, interpolation, input preparation) overlap the next window's GEMMs.

Results per detection are bit-identical to the per-frame path (same per-object seeds, batch-size-independent kernels), so
the ``.mat`` files tools/eval_ycb.py writes do not depend on the window length.
"""
from __future__ import annotations

import numpy as np
import torch

from . import preprocess as pp
from .network import PoseEstimator


class WindowEstimator:
    def __init__(self, estimator, refiner, num_points, iteration, max_frames, frame_hw=(pp.IMG_WIDTH, pp.IMG_LENGTH), cam=pp.YCB_CAM,
                 depth=4):
        self.depth = max(1, int(depth))
        self.num_points, self.iteration, self.cam = int(num_points), int(iteration), cam
        self.dev = next(estimator.parameters()).device
        IH, IW = frame_hw
        self.max_frames = int(max_frames)
        # `depth` windows in flight (callers collect window i - depth + 1 before submitting window i + 1): one device slot,
        # compute stream and workspace each, plus one slot that may be uploading
        self.slots = [dict(rgb=torch.empty(max_frames, IH, IW, 3, dtype=torch.uint8, device=self.dev),
                           depth=torch.empty(max_frames, IH, IW, dtype=torch.int16, device=self.dev),
                           label=torch.empty(max_frames, IH, IW, dtype=torch.int32, device=self.dev),
                           ready=torch.cuda.Event(), done=torch.cuda.Event(), stream=torch.cuda.Stream(device=self.dev),
                           pe=PoseEstimator(estimator, refiner)) for _ in range(self.depth + 1)]
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self._n = 0

    def submit(self, rgb, depth, label, detections):
        """rgb [F,IH,IW,3] uint8, depth [F,IH,IW] int16/uint16 bits, label [F,IH,IW] int32: HOST tensors (pinned for an
        asynchronous upload).  detections: list of (frame, itemid, roi_row, seed) in result order.  Everything is enqueued
        (no host sync); returns a handle for ``collect``."""
        F = rgb.shape[0]
        if F > self.max_frames:
            raise RuntimeError(f"window of {F} frames exceeds max_frames={self.max_frames}")
        slot = self.slots[self._n % len(self.slots)]
        self._n += 1
        main = slot["stream"]
        # several windows in flight: the upload rides on the window's OWN stream (the other windows' compute overlaps it).  A separate
        # copy stream would share one of the runtime's 4 hardware queues with some window's compute stream and every upload would
        # queue behind that window (7 900 -> 8 300 poses/s through this class with 8 queues; with the default 4 this is the fix).
        # One window at a time: a copy stream, so that the next upload overlaps this window's compute.
        up = main if self.depth > 1 else self.copy_stream
        up.wait_stream(torch.cuda.current_stream(self.dev))     # whatever filled the caller's buffers
        with torch.cuda.stream(up):
            slot["rgb"][:F].copy_(rgb, non_blocking=True)
            slot["depth"][:F].copy_(depth.view(torch.int16) if depth.dtype != torch.int16 else depth, non_blocking=True)
            slot["label"][:F].copy_(label, non_blocking=True)
            slot["ready"].record(up)
        main.wait_event(slot["ready"])
        n = len(detections)
        lost = np.zeros(n, dtype=bool)
        buckets = {}
        for k, (frame, itemid, roi, seed) in enumerate(detections):
            bb = pp.get_bbox(roi)
            H, W = bb[1] - bb[0], bb[3] - bb[2]
            if H < 8 or W < 8 or bb[0] < 0 or bb[2] < 0 or bb[1] > rgb.shape[1] or bb[3] > rgb.shape[2]:
                lost[k] = True          # degenerate PoseCNN box: the reference ends in its "Detector Lost" branch (eval_ycb.py:234-237)
                continue
            buckets.setdefault((H, W), []).append((k, frame, int(itemid), bb, int(seed)))
        handle = dict(n=n, lost=lost, order=[], counts=[], out=None, done=slot["done"])
        if not buckets:
            return handle
        imgs, clouds, chooses, objs = [], [], [], []
        with torch.cuda.stream(main):
            for (H, W), members in sorted(buckets.items()):
                objects = [(frame, itemid, bb, seed) for _, frame, itemid, bb, seed in members]
                img, cloud, choose, count = pp.preprocess_objects(slot["rgb"][:F], slot["depth"][:F], slot["label"][:F], objects, self.num_points, self.cam)
                imgs.append(img); clouds.append(cloud); chooses.append(choose.reshape(len(members), -1)); handle["counts"].append(count)
                objs.append(torch.tensor([itemid - 1 for _, _, itemid, _, _ in members], dtype=torch.int64))
                handle["order"] += [k for k, *_ in members]
            obj = torch.cat(objs).pin_memory().to(self.dev, non_blocking=True)
            handle["out"] = slot["pe"].estimate_multi(imgs, torch.cat(clouds), torch.cat(chooses), obj, self.iteration)
            handle["counts"] = torch.cat(handle["counts"])
            handle["host"] = tuple(t.to("cpu", non_blocking=True) for t in (*handle["out"], handle["counts"]))     # pinned by torch
            slot["done"].record(main)
        return handle

    @staticmethod
    def collect(handle):
        """-> (pose_wo_refine [n,7], pose [n,7], lost [n] bool) as numpy, in detection order; lost detections keep zero rows
        (what the reference writes for them)."""
        n = handle["n"]
        wo, ref, lost = np.zeros((n, 7)), np.zeros((n, 7)), handle["lost"].copy()
        if handle["out"] is not None:
            handle["done"].synchronize()
            p_wo, p_ref, counts = (t.numpy() for t in handle["host"])
            for j, k in enumerate(handle["order"]):
                if counts[j] == 0:
                    lost[k] = True
                else:
                    wo[k], ref[k] = p_wo[j], p_ref[j]
        return wo, ref, lost
