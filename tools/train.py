#!/usr/bin/env python
"""Two-phase DenseFusion trainer on the HIP path -- the job of the reference's tools/train.py:51-251.

Phase A trains PoseNet with the confidence-weighted ADD(-S) loss until the test distance drops below
``--refine_margin``; phase B freezes it and trains PoseRefineNet for ``--iteration`` refinement steps per frame.
One frame per forward/backward by default (bs = 1, like the reference; ``--frames_per_pass P`` lets up to P frames of
equal crop size out of one accumulation window share a pass), ``--batch_size`` frames accumulated per optimizer step, Adam, per-epoch test pass, ``*_current.pth`` every 1000 frames and best-model checkpoints with the
reference's file names, so its eval scripts and ours load them.

Data-parallel over the GPUs of one node (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL over
xGMI): every rank trains on its own shard of the frame list, gradients live in ONE flat fp32 buffer
(85.8 MB for PoseNet, 7.8 MB for the refiner) that is summed with ONE all-reduce per optimizer step; nothing
else is communicated.  Launch: ``python -m torch.distributed.run --nproc-per-node N tools/train.py ...``.

``--dataset synthetic`` trains on seeded synthetic frames (no dataset ships offline); ``ycb`` / ``linemod`` use the built-in
loaders (``densefusion_amd.datasets``: the reference's constructor, file layout, 6-tuple and training augmentation -- colour jitter,
occluders, synthetic frames over real backgrounds, pose-translation noise -- with the crop / sampling / back-projection on the
device and the decode in worker processes); ``--reference_loaders`` imports ``datasets.<name>.dataset.PoseDataset`` from the
PYTHONPATH instead (the reference's loaders work unchanged: they return the same 6-tuple).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densefusion_amd import synth, train_ops, train_utils  # noqa: E402
from densefusion_amd.lib.loss import Loss  # noqa: E402
from densefusion_amd.lib.loss_refiner import Loss_refine  # noqa: E402
from densefusion_amd.lib.network import PoseNet, PoseRefineNet  # noqa: E402
from densefusion_amd.native_train import NativeTrainer  # noqa: E402


class SyntheticPoseDataset(torch.utils.data.Dataset):
    """Seeded synthetic frames in the reference's dataset tuple layout (datasets/ycb/dataset.py:227-232)."""
    CROPS = [(80, 80), (120, 120), (120, 160), (160, 160)]

    def __init__(self, mode, num_pt, num_obj, length, num_pt_mesh=500, sym=(12, 15, 18, 19, 20), crops=None, seed=0):
        self.mode, self.num_pt, self.num_obj, self.length = mode, num_pt, num_obj, length
        self.num_pt_mesh, self.sym = num_pt_mesh, [s for s in sym if s < num_obj]
        self.crops = crops or self.CROPS
        self.seed = seed + (0 if mode == "train" else 10_000_000)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        H, W = self.crops[i % len(self.crops)]
        o = synth.make_object(self.seed + i, H, W, self.num_pt, self.num_obj, self.num_pt_mesh)
        return (torch.from_numpy(o["cloud"]), torch.from_numpy(o["choose"]), torch.from_numpy(o["img"]),
                torch.from_numpy(o["target"]), torch.from_numpy(o["model_points"]), torch.from_numpy(o["obj"]))

    def get_sym_list(self):
        return self.sym

    def get_num_points_mesh(self):
        return self.num_pt_mesh


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str, default="synthetic", help="synthetic | ycb | linemod")
    ap.add_argument("--dataset_root", type=str, default="")
    ap.add_argument("--dataset_config_dir", type=str, default="datasets/ycb/dataset_config",
                    help="ycb: the directory of classes.txt / train_data_list.txt / test_data_list.txt (the reference keeps them in its tree)")
    ap.add_argument("--batch_size", type=int, default=32, help="frames accumulated per optimizer step and GPU (this fork's default, tools/train.py:34 of the reference; BASELINE configs[3] runs with --batch_size 8)")
    ap.add_argument("--frames_per_pass", type=int, default=1,
                    help="frames of equal crop size, out of one accumulation window, that share a forward/backward pass (1 = the "
                         "reference's bs = 1 passes; the gradients of a window are the same either way)")
    ap.add_argument("--workers", type=int, default=4, help="prefetch workers: frames are fetched, pinned and uploaded on a copy stream "
                                                         "ahead of the step (0 = fetch in the training loop)")
    ap.add_argument("--feed", type=str, default="processes", choices=["processes", "threads"],
                    help="processes: the disk datasets decode in --workers worker processes (the reference's DataLoader workers) and the "
                         "trainer's process only uploads; threads: --workers threads of the trainer's process do both (always for the synthetic set)")
    ap.add_argument("--reference_loaders", action="store_true",
                    help="import datasets.<name>.dataset.PoseDataset from the PYTHONPATH (the reference's loaders) instead of the built-in ones")
    ap.add_argument("--lanes", type=int, default=4,
                    help="PoseNet phase, native step: passes of one accumulation window run on this many concurrent lanes (own HIP stream, "
                         "host thread, workspace and gradient buffer each; gradients summed in lane order): bs = 1 passes fill a fraction "
                         "of the chip, so independent frames overlap.  1 = one pass at a time")
    ap.add_argument("--passes", type=str, default="window", choices=["window", "lanes"],
                    help="native step.  window (default): the frames of an accumulation window, whatever their crop sizes, run as ONE "
                         "multi-bucket pass (df_posenet_train_step_multi: per-point layers, 1x1 / Winograd-domain products and every weight gradient "
                         "once over all frames; in the refiner phase: df_posenet_forward_multi for the frozen estimator, then the refiner steps over "
                         "all frames of the window); lanes: one pass per --frames_per_pass frames of equal size, spread over --lanes concurrent lanes")
    ap.add_argument("--window_pixels", type=int, default=1 << 21,
                    help="--passes window: a window whose crops add up to more pixels than this is cut into several passes (bounds the workspace: "
                         "about 4.6 KB per crop pixel)")
    ap.add_argument("--autograd_tape", action="store_true",
                    help="train through the per-layer autograd Functions of round 2 (lib/train_graph.py) instead of the native step "
                         "(csrc/train.hip: forward + loss + backward of a pass in one library call); same gradients, several times slower")
    # optimisation defaults: THIS fork's (tools/train.py:34-42 of the reference), not upstream DenseFusion's
    ap.add_argument("--lr", type=float, default=0.0001)
    ap.add_argument("--lr_rate", type=float, default=0.1)
    ap.add_argument("--w", type=float, default=0.015)
    ap.add_argument("--w_rate", type=float, default=0.1)
    ap.add_argument("--decay_margin", type=float, default=0.03)
    ap.add_argument("--refine_margin", type=float, default=0.02)
    ap.add_argument("--noise_trans", type=float, default=0.005)
    ap.add_argument("--iteration", type=int, default=2)
    ap.add_argument("--nepoch", type=int, default=500)
    ap.add_argument("--repeat_epoch", type=int, default=0,
                    help="passes over the training set per epoch, i.e. per test pass / decay check / refine check (tools/train.py:61-73,143 of the "
                         "reference: 20 for linemod, 1 otherwise); 0 = that rule")
    ap.add_argument("--resume_posenet", type=str, default="")
    ap.add_argument("--resume_refinenet", type=str, default="")
    ap.add_argument("--start_epoch", type=int, default=1)
    ap.add_argument("--outf", type=str, default="", help="default trained_models/<dataset> (tools/train.py:58-66 of the reference)")
    ap.add_argument("--log_dir", type=str, default="", help="default experiments/logs/<dataset>")
    # synthetic-data knobs
    ap.add_argument("--num_objects", type=int, default=21)
    ap.add_argument("--num_points", type=int, default=1000)
    ap.add_argument("--synthetic_train_frames", type=int, default=64)
    ap.add_argument("--synthetic_test_frames", type=int, default=16)
    ap.add_argument("--refine_start", action="store_true")
    ap.add_argument("--seed", type=int, default=1000, help="base seed; the epoch permutation uses seed + epoch on EVERY rank")
    ap.add_argument("--dist_backend", type=str, default="nccl", help="torch.distributed backend under torch.distributed.run "
                                                                     "(nccl = RCCL over xGMI; gloo to rehearse several ranks on one GPU)")
    return ap


def make_datasets(opt):
    if opt.dataset == "synthetic":
        tr = SyntheticPoseDataset("train", opt.num_points, opt.num_objects, opt.synthetic_train_frames)
        te = SyntheticPoseDataset("test", opt.num_points, opt.num_objects, opt.synthetic_test_frames)
        return tr, te
    if opt.dataset not in ("ycb", "linemod"):
        raise SystemExit("Unknown dataset")
    opt.num_objects, opt.num_points = (21, 1000) if opt.dataset == "ycb" else (13, 500)
    if opt.reference_loaders:        # the reference's own loaders from the PYTHONPATH (host-side preparation, its DataLoader semantics)
        PoseDataset = __import__("datasets.%s.dataset" % opt.dataset, fromlist=["PoseDataset"]).PoseDataset
        return (PoseDataset("train", opt.num_points, True, opt.dataset_root, opt.noise_trans, opt.refine_start),
                PoseDataset("test", opt.num_points, False, opt.dataset_root, 0.0, opt.refine_start))
    # built-in loaders: the reference's constructor arguments (tools/train.py:57-66: augmentation on for the training set)
    if opt.dataset == "ycb":
        from densefusion_amd.datasets.ycb.dataset import PoseDataset
    else:
        from densefusion_amd.datasets.linemod.dataset import PoseDataset
    logging.getLogger("train").info("datasets.%s: the built-in loader (device-side preparation; training augmentation on, noise_trans %g)",
                                    opt.dataset, opt.noise_trans)
    kw = dict(dataset_config_dir=opt.dataset_config_dir) if opt.dataset == "ycb" else {}
    return (PoseDataset("train", opt.num_points, True, opt.dataset_root, opt.noise_trans, opt.refine_start, **kw),
            PoseDataset("test", opt.num_points, False, opt.dataset_root, 0.0, opt.refine_start, **kw))


def main(argv=None):
    opt = build_parser().parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if opt.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(opt.dist_backend, rank=rank, world_size=world)
    if "DF_TRAIN_DEVICE" in os.environ:            # rehearsal only: several ranks sharing one card
        local = int(os.environ["DF_TRAIN_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(opt.seed + rank)             # dropout masks / initial weights may differ per rank; the data order may not
    np.random.seed(opt.seed + rank)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(message)s")
    log = logging.getLogger("train")
    opt.outf = opt.outf or "trained_models/%s" % opt.dataset
    opt.log_dir = opt.log_dir or "experiments/logs/%s" % opt.dataset
    os.makedirs(opt.outf, exist_ok=True)
    os.makedirs(opt.log_dir, exist_ok=True)

    decay_start = False
    if opt.resume_refinenet:
        # a resumed refiner run IS the refiner phase (tools/train.py:86-100): set it BEFORE the datasets are built (YCB
        # samples 2600 mesh points instead of 500 when refine = True, datasets/ycb/dataset.py:90-91,240-244), and the
        # learning-rate / w decay has already been applied -- it must not be applied a second time
        opt.refine_start = True
        decay_start = True
        opt.lr *= opt.lr_rate
        opt.w *= opt.w_rate
        opt.batch_size = max(1, int(opt.batch_size / opt.iteration))
    if opt.repeat_epoch <= 0:
        opt.repeat_epoch = 20 if opt.dataset == "linemod" else 1
    dataset, test_dataset = make_datasets(opt)
    estimator = PoseNet(num_points=opt.num_points, num_obj=opt.num_objects).to(dev)
    refiner = PoseRefineNet(num_points=opt.num_points, num_obj=opt.num_objects).to(dev)
    if world > 1:
        # data-parallel replicas must start from the same weights: rank 0's initialisation goes to everybody, decided PER
        # NETWORK -- a run resumed from a PoseNet checkpoint alone still builds a fresh refiner on every rank
        fresh = ([] if opt.resume_posenet else list(estimator.parameters())) + ([] if opt.resume_refinenet else list(refiner.parameters()))
        for prm in fresh:
            buf = prm.data if opt.dist_backend == "nccl" else prm.data.cpu()
            dist.broadcast(buf, 0)
            if opt.dist_backend != "nccl":
                prm.data.copy_(buf)
    if opt.resume_posenet:
        estimator.load_state_dict(torch.load(os.path.join(opt.outf, opt.resume_posenet), map_location=dev, weights_only=True))
    if opt.resume_refinenet:
        refiner.load_state_dict(torch.load(os.path.join(opt.outf, opt.resume_refinenet), map_location=dev, weights_only=True))

    if not train_utils.replicas_in_sync([estimator, refiner]):
        raise RuntimeError("data-parallel ranks hold different weights after initialisation / resume")

    # Native step (default): parameters and gradients of the network being trained live in the library's flat kernel-layout
    # buffers (NativeTrainer); the nn.Modules keep serving the inference engine (frozen estimator of the refiner phase, test
    # passes) and are refreshed from the flat buffer before they are used / saved.
    native = {} if opt.autograd_tape else {"posenet": NativeTrainer("posenet", opt.num_points, opt.num_objects, dev),
                                           "refiner": NativeTrainer("refiner", opt.num_points, opt.num_objects, dev)}
    if native:
        native["posenet"].load_state_dict(estimator.state_dict())
        native["refiner"].load_state_dict(refiner.state_dict())

    lanes = None
    if native and opt.lanes > 1 and opt.passes == "lanes":
        from densefusion_amd.native_train import Lanes
        lanes = Lanes(native["posenet"], opt.lanes)

    refine_lanes = []          # [Lanes] once the refiner phase has started

    def start_refine_lanes():
        """Refiner phase on lanes: every lane gets its own refiner step (workspace, gradient buffer) AND its own copy of the frozen
        estimator (engine handle + workspace are per module), so the frames of a window run side by side like the PoseNet phase's."""
        if native and not refine_lanes:
            sync_module(estimator)
        if not native or opt.lanes <= 1 or refine_lanes or opt.passes != "lanes":
            return
        from densefusion_amd.native_train import Lanes
        rl = Lanes(native["refiner"], opt.lanes)
        rl.lanes[0].frozen_estimator = estimator
        for lane in rl.lanes[1:]:
            e = PoseNet(num_points=opt.num_points, num_obj=opt.num_objects)
            e.load_state_dict(estimator.state_dict())
            lane.frozen_estimator = e.to(dev).eval()
        refine_lanes.append(rl)

    def optimizer_for(module):
        flat = native["posenet" if module is estimator else "refiner"] if native else train_utils.FlatParams(module)
        return flat, train_utils.FlatAdam(flat, lr=opt.lr)

    def sync_module(module):
        """the nn.Module's tensors <- the flat buffer the native step trains (no-op on the autograd-tape path: same storage)"""
        if native:
            module.load_state_dict(native["posenet" if module is estimator else "refiner"].state_dict())
        return module

    flat, optimizer = optimizer_for(refiner if opt.refine_start else estimator)
    if opt.refine_start:
        start_refine_lanes()
    opt.sym_list = dataset.get_sym_list()
    opt.num_points_mesh = dataset.get_num_points_mesh()
    criterion = Loss(opt.num_points_mesh, opt.sym_list)
    criterion_refine = Loss_refine(opt.num_points_mesh, opt.sym_list)
    log.info(">>>>>>>>----------Dataset loaded!---------<<<<<<<<\nlength of the training set: %d\nlength of the testing set: %d\n"
             "number of sample points on mesh: %d\nsymmetry object list: %s", len(dataset), len(test_dataset), opt.num_points_mesh, opt.sym_list)

    def allreduce(fl):
        if world > 1 and opt.dist_backend != "nccl":              # gloo rehearsal: through host memory
            g = fl.grad.cpu()
            dist.all_reduce(g)
            fl.grad.copy_(g)
            return world
        return train_utils.allreduce_gradients(fl)

    def to_dev(data):
        points, choose, img, target, model_points, idx = data
        if points.dim() == 1:                       # the LineMOD loader's "lost detection" sentinel: six LongTensor([0])
            return None
        on = lambda t: t if t.device == dev else t.to(dev)        # (the prefetcher hands over device tensors: no dispatch for a no-op)
        f = lambda t: on(t)[None]                    # add the bs = 1 axis the DataLoader of the reference adds
        return (f(points), on(choose).reshape(1, 1, -1), f(img), f(target), f(model_points),
                train_utils.with_host_index(on(idx).reshape(1, 1), getattr(idx, "_host", idx)))   # host copy kept: no read-back

    def run_pass(frames):
        """Forward + loss + backward of `frames` (same crop size) in one pass; returns their distances (a device tensor [len(frames)]:
        nothing is read back inside the accumulation window)."""
        if native:
            return _run_pass_native(frames)
        with train_ops.splitk_scope(dev):                # the small-map convolutions split their reductions (one registration per pass)
            return torch.tensor(_run_pass(frames), device=dev)

    def _run_pass_native(frames, lane=None):
        points, choose, img, target, model_points = (torch.cat([f[k] for f in frames]) for k in (0, 1, 2, 3, 4))
        idx = torch.cat([f[5] for f in frames])
        sym = [train_utils.host_index(f[5]) in opt.sym_list for f in frames]
        if not opt.refine_start:
            return (lane or native["posenet"]).step_posenet(img, points, choose, idx, target, model_points, sym, opt.w, dropout=True)["dis"]
        with torch.no_grad():                            # the frozen estimator of the refiner phase: the fused inference engine
            pred_r, pred_t, pred_c, emb = getattr(lane, "frozen_estimator", estimator)(img, points, choose, idx)
            new_points, new_target = [], []
            for b, f in enumerate(frames):
                _, _, npt, ntg = criterion(pred_r[b:b + 1], pred_t[b:b + 1], pred_c[b:b + 1], f[3], f[4], f[5], f[0], opt.w, True)
                new_points.append(npt); new_target.append(ntg)
            new_points, new_target = torch.cat(new_points), torch.cat(new_target)
        for _ in range(opt.iteration):
            out = (lane or native["refiner"]).step_refiner(new_points, emb, idx, new_target, model_points, sym)
            new_points, new_target = out["new_points"], out["new_target"]
        return out["dis"]

    def _pixel_chunks(frames, limit):
        """`frames` cut into runs of at most `limit` crop pixels (at least one frame each)"""
        chunks, cur, px = [], [], 0
        for f in frames:
            n = int(f[2].shape[-2]) * int(f[2].shape[-1])
            if cur and px + n > limit:
                chunks.append(cur); cur, px = [], 0
            cur.append(f); px += n
        if cur:
            chunks.append(cur)
        return chunks

    def _run_window_native(frames):
        """PoseNet phase: the frames of a window (any crop sizes) as multi-bucket passes of at most --window_pixels crop pixels each; returns
        the frames' distances (device tensor)."""
        dists = []
        for chunk in _pixel_chunks(frames, opt.window_pixels):
            by_size = {}
            for f in chunk:
                by_size.setdefault(tuple(f[2].shape[-2:]), []).append(f)
            order = [f for group in by_size.values() for f in group]
            imgs = [torch.cat([f[2] for f in group]) for group in by_size.values()]
            cat = lambda k: torch.cat([f[k] for f in order])
            sym = [train_utils.host_index(f[5]) in opt.sym_list for f in order]
            dists.append(native["posenet"].step_posenet_multi(imgs, cat(0), cat(1), cat(5), cat(3), cat(4), sym, opt.w, dropout=True)["dis"])
        return torch.cat(dists)

    def _run_window_refine_native(frames):
        """Refiner phase (tools/train.py:139-159) on a whole window: the frozen estimator over all crop sizes in multi-bucket forwards of the
        inference engine (at most 2 x --window_pixels crop pixels each: the engine keeps no activations), the arg-max-confidence re-centring per
        frame (Loss with refine=True), then `iteration` native refiner steps over ALL frames of the window at once (the refiner sees num_points
        points per frame whatever the crop size)."""
        order, emb, new_points, new_target = [], [], [], []
        with torch.no_grad():
            for chunk in _pixel_chunks(frames, 2 * opt.window_pixels):
                by_size = {}
                for f in chunk:
                    by_size.setdefault(tuple(f[2].shape[-2:]), []).append(f)
                part = [f for group in by_size.values() for f in group]
                cat = lambda k: torch.cat([f[k] for f in part])
                pred_r, pred_t, pred_c, e = estimator.forward_multi([torch.cat([f[2] for f in group]) for group in by_size.values()], cat(0), cat(1), cat(5))
                _, _, npt, ntg = criterion.forward_frames(pred_r, pred_t, pred_c, cat(3), cat(4), [train_utils.host_index(f[5]) for f in part], cat(0),
                                                          opt.w, True)           # lib/loss.py per frame, all frames of the chunk in one call
                new_points.append(npt); new_target.append(ntg)
                order += part; emb.append(e)
            new_points, new_target, emb = torch.cat(new_points), torch.cat(new_target), torch.cat(emb)
        idx, model_points = torch.cat([f[5] for f in order]), torch.cat([f[4] for f in order])
        sym = [train_utils.host_index(f[5]) in opt.sym_list for f in order]
        for _ in range(opt.iteration):
            out = native["refiner"].step_refiner(new_points, emb, idx, new_target, model_points, sym)
            new_points, new_target = out["new_points"], out["new_target"]
        return out["dis"]

    def _run_pass(frames):
        points, choose, img = (torch.cat([f[k] for f in frames]) for k in (0, 1, 2))
        idx = torch.cat([f[5] for f in frames])
        if opt.refine_start:
            with torch.no_grad():
                pred_r, pred_t, pred_c, emb = estimator(img, points, choose, idx)
                new_points, new_target = [], []
                for b, f in enumerate(frames):
                    _, dis, npt, ntg = criterion(pred_r[b:b + 1], pred_t[b:b + 1], pred_c[b:b + 1], f[3], f[4], f[5], f[0], opt.w, True)
                    new_points.append(npt); new_target.append(ntg)
            last = [None] * len(frames)
            for _ in range(opt.iteration):
                pr, pt = refiner(torch.cat(new_points), emb, idx)
                total = 0
                for b, f in enumerate(frames):
                    dis, new_points[b], new_target[b] = criterion_refine(pr[b:b + 1], pt[b:b + 1], new_target[b], f[4], f[5], new_points[b])
                    total = total + dis
                    last[b] = dis
                total.backward()
            return [float(d.detach()) for d in last]
        pred_r, pred_t, pred_c, emb = estimator(img, points, choose, idx)
        total, dists = 0, []
        for b, f in enumerate(frames):
            loss, dis, _, _ = criterion(pred_r[b:b + 1], pred_t[b:b + 1], pred_c[b:b + 1], f[3], f[4], f[5], f[0], opt.w, False)
            total = total + loss
            dists.append(dis)
        total.backward()
        return [float(d.detach()) for d in dists]

    dis_host = [torch.zeros((), dtype=torch.float32).pin_memory() for _ in range(2)]
    pending_log = []

    def flush_log():
        while pending_log:
            ev, slot, ep, batch_no, count, bs = pending_log.pop(0)
            ev.synchronize()
            log.info("Train time %s Epoch %d Batch %d Frame %d Avg_dis:%f", time.strftime("%Hh %Mm %Ss", time.gmtime(time.time() - st_time)),
                     ep, batch_no, count, float(slot) / bs)

    feeds = {}

    def feed(ds, order):
        """The look-ahead over `ds` in `order`; one Prefetcher per dataset, so its worker processes persist over the epochs."""
        pf = feeds.get(id(ds))
        if pf is None:
            pf = feeds[id(ds)] = train_utils.Prefetcher(ds, order, dev, workers=opt.workers,
                                                        processes=opt.workers if opt.feed == "processes" else 0)
        return pf.set_order(order)

    def close_feeds():
        for pf in feeds.values():
            pf.close()
        feeds.clear()

    best_test = np.inf
    st_time = time.time()
    frames_seen = 0
    def log_file(name):
        """Per-epoch log files like the reference's setup_logger (lib/utils.py:3-17; tools/train.py:132,182): rank 0 only."""
        if rank != 0:
            return None
        h = logging.FileHandler(os.path.join(opt.log_dir, name), mode="w")
        h.setFormatter(logging.Formatter("%(asctime)s : %(message)s"))
        log.addHandler(h)
        return h

    def close_log(h):
        if h is not None:
            log.removeHandler(h)
            h.close()

    for epoch in range(opt.start_epoch, opt.nepoch):
        fh = log_file("epoch_%d_log.txt" % epoch)
        if opt.refine_start:
            estimator.eval(); refiner.train()
        else:
            estimator.train()
        flat.zero_grad()
        train_count = 0
        # this rank's shard of the epoch.  The permutation comes from a seed EVERY rank shares (seed + epoch), so the shards are
        # disjoint; all ranks take the same number of frames and therefore the same number of optimizer steps -- the gradient
        # all-reduce is a collective, a rank that ran one step fewer would leave the others waiting in it.  Lost-detection
        # sentinels (LineMOD) count toward the window like any frame: they add no gradient but never skip a collective.
        steps = len(dataset) // (world * opt.batch_size)
        if steps == 0:
            log.warning("epoch %d: %d training frames are fewer than ranks x batch_size = %d: no optimizer step this epoch", epoch, len(dataset),
                        world * opt.batch_size)
        # `repeat_epoch` passes over the set per epoch, each with its own permutation (seed, epoch, pass) shared by every rank
        order = np.concatenate([np.random.RandomState((opt.seed + epoch) * 1009 + rep).permutation(len(dataset))[:steps * world * opt.batch_size][rank::world]
                                for rep in range(opt.repeat_epoch)]) if steps else np.zeros(0, dtype=np.int64)
        window, slots = [], 0
        window_dis = torch.zeros((), device=dev)
        for item in feed(dataset, order):
            data = to_dev(item)
            slots += 1
            if data is not None:
                window.append(data)
            if slots < opt.batch_size:
                continue
            slots = 0
            # one accumulation window = one optimizer step (tools/train.py:131-170); frames of equal crop size may share
            # a pass (--frames_per_pass): the summed gradient of the window does not depend on how it is cut into passes
            by_size = {}
            for f in window:
                by_size.setdefault(tuple(f[2].shape[-2:]), []).append(f)
            passes = [group[g0:g0 + max(1, opt.frames_per_pass)] for group in by_size.values() for g0 in range(0, len(group), max(1, opt.frames_per_pass))]
            active = (refine_lanes[0] if refine_lanes else None) if opt.refine_start else lanes
            if native and opt.passes == "window":
                if window:
                    window_dis = window_dis + (_run_window_refine_native if opt.refine_start else _run_window_native)(window).sum()
            elif active is not None:
                for d in active.run([(lambda lane, fs=fs: _run_pass_native(fs, lane)) for fs in passes]):
                    window_dis = window_dis + d.sum()
            else:
                for fs in passes:
                    window_dis = window_dis + run_pass(fs).sum()
            window = []
            prev = train_count
            train_count += opt.batch_size
            frames_seen += opt.batch_size
            n = allreduce(flat)                                   # the one collective of the training path
            optimizer.step(grad_scale=1.0 / n)
            flat.zero_grad()
            # the window's one read-back (its log line needs the number) trails the launches by one window: an asynchronous copy into
            # pinned memory now, waited for only after the NEXT window has been enqueued -- the device never idles on the host
            slot = dis_host[(train_count // opt.batch_size) % 2]
            slot.copy_(window_dis, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            window_dis = torch.zeros((), device=dev)
            flush_log()
            pending_log.append((ev, slot, epoch, train_count // opt.batch_size, train_count, opt.batch_size))
            if train_count // 1000 != prev // 1000 and rank == 0:
                if opt.refine_start:
                    torch.save(sync_module(refiner).state_dict(), "{0}/pose_refine_model_current.pth".format(opt.outf))
                else:
                    torch.save(sync_module(estimator).state_dict(), "{0}/pose_model_current.pth".format(opt.outf))
        flush_log()
        log.info(">>>>>>>>----------epoch %d train finish---------<<<<<<<<", epoch)
        close_log(fh)
        fh = log_file("epoch_%d_test_log.txt" % epoch)

        # per-epoch test pass (tools/train.py:181-209): the fused inference engine, no gradients
        sync_module(refiner if opt.refine_start else estimator)
        estimator.eval(); refiner.eval()
        test_dis, test_count = torch.zeros((), dtype=torch.float64, device=dev), 0          # summed on the device: one read-back per test pass
        with torch.no_grad():
            for item in feed(test_dataset, range(rank, len(test_dataset), world)):
                data = to_dev(item)
                if data is None:
                    continue
                points, choose, img, target, model_points, idx = data
                pred_r, pred_t, pred_c, emb = estimator(img, points, choose, idx)
                _, dis, new_points, new_target = criterion(pred_r, pred_t, pred_c, target, model_points, idx, points, opt.w, opt.refine_start)
                if opt.refine_start:
                    for _ in range(opt.iteration):
                        pred_r, pred_t = refiner(new_points, emb, idx)
                        dis, new_points, new_target = criterion_refine(pred_r, pred_t, new_target, model_points, idx, new_points)
                test_dis = test_dis + dis.reshape(()).double()
                test_count += 1
        stats = torch.tensor([float(test_dis), float(test_count)], device=dev if opt.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(stats)
        test_dis = float(stats[0] / max(stats[1], 1.0))
        log.info("Test time %s Epoch %d TEST FINISH Avg dis: %f", time.strftime("%Hh %Mm %Ss", time.gmtime(time.time() - st_time)), epoch, test_dis)
        close_log(fh)
        if test_dis <= best_test:
            best_test = test_dis
            if rank == 0:
                if opt.refine_start:
                    torch.save(refiner.state_dict(), "{0}/pose_refine_model_{1}_{2}.pth".format(opt.outf, epoch, test_dis))
                else:
                    torch.save(estimator.state_dict(), "{0}/pose_model_{1}_{2}.pth".format(opt.outf, epoch, test_dis))
                log.info("%d >>>>>>>>----------BEST TEST MODEL SAVED---------<<<<<<<<", epoch)
        if best_test < opt.decay_margin and not decay_start:
            decay_start = True
            opt.lr *= opt.lr_rate
            opt.w *= opt.w_rate
            optimizer = train_utils.FlatAdam(flat, lr=opt.lr)
            log.info("decay: lr -> %g, w -> %g", opt.lr, opt.w)
        if best_test < opt.refine_margin and not opt.refine_start:
            opt.refine_start = True
            opt.batch_size = max(1, int(opt.batch_size / opt.iteration))
            flat, optimizer = optimizer_for(refiner)
            start_refine_lanes()
            if opt.dataset != "synthetic":
                close_feeds()
                dataset, test_dataset = make_datasets(opt)
                opt.sym_list, opt.num_points_mesh = dataset.get_sym_list(), dataset.get_num_points_mesh()
                criterion, criterion_refine = Loss(opt.num_points_mesh, opt.sym_list), Loss_refine(opt.num_points_mesh, opt.sym_list)
    close_feeds()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return best_test


if __name__ == "__main__":
    main()
