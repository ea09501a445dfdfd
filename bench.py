#!/usr/bin/env python
"""bench.py -- poses/sec of the DenseFusion hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (PoseNet -> per-pixel pose selection -> 2 refine
iterations, all on the device) over one batch of synthetic (frame, object) samples already resident
in HBM: the YCB-Video-shaped stream of SURVEY 8d config 3 -- K=21 objects, N=1000 points, crops cycled
over seven snapped sizes, `--per-bucket` objects of each size per step (5 objects/frame).  Objects of
one crop size are evaluated as one batched launch sequence; the step is captured in a hipGraph.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (weak scaling:
        every rank owns its own batch; no collective on the data path, one tiny all_gather of the
        [n,7] poses per step)

Prints ONE JSON line (rank 0).  Extra objects in it:
  roofline     -- the implicit-GEMM fp32-MFMA kernel (igemm_f32_kernel, >90 % of the step): algorithmic
                  FLOPs / summed launch durations, durations taken with HIP events on the launch stream
                  in an instrumented re-run of the same steps (events between launches would perturb
                  the timed region itself)
  knn          -- the fused 1-NN kernel at the YCB symmetric-loss size (R=500, Q=500 000)
  cpu_baseline -- the CPU oracle (a port of the reference path) timed on this host, bounded sample
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, the
    environment torch.distributed.run would give them), relay rank 0's JSON line and exit with the worst child status.  Runs
    before torch or the HIP library is imported: the parent never touches the GPU and replaces no process (no exec)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # supervise every rank: rank 0's output is drained by a thread while all children are polled; the first rank that fails (or
    # the overall deadline) ends the others -- a dead rank would otherwise leave the rest in a collective until the RCCL timeout
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("DF_BENCH_DEADLINE_S", "1500"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} exited with status {rcs[bad[0]]}"
        elif time.time() > deadline:
            failed = "deadline exceeded"
        if failed or all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    out = got[0] if got else ""
    if failed:
        print(f"[bench] {failed}; rank exit codes {rcs}", file=sys.stderr)
        for ln in (out or "").splitlines():
            print(ln, file=sys.stderr)
        first = next((rc for rc in rcs if rc), 1)
        sys.exit(first if first > 0 else 1)
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is None or json.loads(line).get("n_gpus") != n:
        print(f"[bench] rank 0 did not report n_gpus == {n}: {line}", file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)
    sys.exit(0)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _pre = argparse.ArgumentParser(add_help=False)
    _pre.add_argument("--gpus", type=int, default=1)
    _n = _pre.parse_known_args()[0].gpus
    if _n > 1:
        spawn_ranks(_n, sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

from densefusion_amd import _lib, sharding, synth  # noqa: E402
from densefusion_amd.lib.knn import KNearestNeighbor  # noqa: E402
from densefusion_amd.lib.network import PoseEstimator, PoseNet, PoseRefineNet  # noqa: E402

CROPS = [(80, 80), (120, 120), (120, 160), (160, 160), (160, 200), (200, 240), (240, 320)]   # SURVEY 8d cfg 3
K_OBJ, N_PTS, ITERS = 21, 1000, 2
WSEED = 13
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA = vector peak
HBM_PEAK_GBS = 8000.0


def algorithmic_gflop_per_pose(H, W):
    # 0.896 MFLOP x H*W (CNN) + 7.98 MFLOP x N (point MLPs + heads) + iters x 1.481 MFLOP x N
    return (0.896e6 * H * W + 7.98e6 * N_PTS + ITERS * 1.481e6 * N_PTS) / 1e9


def load_nets(device):
    est, ref = PoseNet(N_PTS, K_OBJ), PoseRefineNet(N_PTS, K_OBJ)
    est.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K_OBJ), WSEED).items()})
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K_OBJ), WSEED + 1000).items()})
    return est.to(device).eval(), ref.to(device).eval()


def make_buckets(rank, world, per_bucket, device):
    """Global stream: object i has crop CROPS[i % 7] and seed 3000+i; rank r evaluates its shard_plan part."""
    total = per_bucket * len(CROPS) * world
    sizes = [CROPS[i % len(CROPS)] for i in range(total)]
    plan = sharding.shard_plan(sizes, world, rank)
    buckets = []
    for (H, W), idxs in plan.items():
        assert len(idxs) == per_bucket
        objs = [synth.make_object(3000 + i, H, W, N_PTS, K_OBJ) for i in idxs]
        b = {k: np.stack([o[k] for o in objs]) for k in objs[0]}
        buckets.append(dict(H=H, W=W, host=b, idx=idxs,
                            img=torch.from_numpy(b["img"]).to(device), cloud=torch.from_numpy(b["cloud"]).to(device),
                            choose=torch.from_numpy(b["choose"]).to(device), obj=torch.from_numpy(b["obj"]).to(device),
                            out=(torch.empty(per_bucket, 7, dtype=torch.float64, device=device),
                                 torch.empty(per_bucket, 7, dtype=torch.float64, device=device))))
    return buckets


def make_groups(buckets, G, device):
    """Split every crop-size bucket into G equal parts; group g = part g of every bucket = one df_estimate_poses_multi call
    (objects concatenated in bucket order).  G = 0: one group per crop size (the per-bucket launch sequences of round 1)."""
    groups = []
    if G <= 0:
        parts = [[(bi, 0, b["img"].shape[0])] for bi, b in enumerate(buckets)]
    else:
        parts = []
        for g in range(G):
            part = []
            for bi, b in enumerate(buckets):
                n = b["img"].shape[0]
                lo, hi = g * n // G, (g + 1) * n // G
                if hi > lo:
                    part.append((bi, lo, hi))
            parts.append(part)
    for part in parts:
        cat = lambda k: torch.cat([buckets[bi][k][lo:hi] for bi, lo, hi in part]).contiguous()
        n = sum(hi - lo for _, lo, hi in part)
        groups.append(dict(part=part, imgs=[buckets[bi]["img"][lo:hi] for bi, lo, hi in part], cloud=cat("cloud"),
                           choose=cat("choose").reshape(n, -1), obj=cat("obj").reshape(n),
                           out=(torch.empty(n, 7, dtype=torch.float64, device=device), torch.empty(n, 7, dtype=torch.float64, device=device))))
    return groups


def run_step(pe, groups, streams=None):
    """One pass over all groups.  With `streams`, every group runs on its own HIP stream (forked from and joined back into the
    current stream) so that one group's memory-bound glue kernels overlap with another's GEMMs; each group has its own
    workspace, i.e. its own PoseEstimator."""
    if streams is None:
        for i, g in enumerate(groups):
            pe[i].estimate_multi(g["imgs"], g["cloud"], g["choose"], g["obj"], ITERS, out=g["out"])
        return
    main = torch.cuda.current_stream()
    for i in reversed(range(len(groups))):
        g, st = groups[i], streams[i]
        st.wait_stream(main)
        with torch.cuda.stream(st):
            pe[i].estimate_multi(g["imgs"], g["cloud"], g["choose"], g["obj"], ITERS, out=g["out"])
    for st in streams:
        main.wait_stream(st)


def bucket_poses(buckets, groups):
    """final poses per crop-size bucket (host), reassembled from the groups' outputs"""
    out = [np.zeros((b["img"].shape[0], 7)) for b in buckets]
    for g in groups:
        p = g["out"][1].cpu().numpy()
        o = 0
        for bi, lo, hi in g["part"]:
            out[bi][lo:hi] = p[o:o + hi - lo]
            o += hi - lo
    return out


def profile_gemm(pe, groups, steps):
    """Instrumented SERIAL re-run (one stream, no graph): HIP events around every GEMM launch on the launch stream."""
    L = _lib.lib()
    hp, hr = pe[0].estimator._handle, pe[0].refiner._handle
    L.df_net_profile(hp, 1); L.df_net_profile(hr, 1)
    tot = np.zeros(4)
    tot_n = 0
    for _ in range(steps):
        run_step(pe, groups)
        torch.cuda.synchronize()
        for h in (hp, hr):
            ms, fl, us, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
            _lib.check(L.df_net_profile_read(h, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(us), ctypes.byref(by), ctypes.byref(n)), "profile_read")
            tot += (ms.value, fl.value, us.value, by.value); tot_n += n.value
    L.df_net_profile(hp, 0); L.df_net_profile(hr, 0)
    return tot[0], tot[1], tot[2], tot[3], tot_n


TRAFFIC_FILE = "r04_igemm_traffic.json"


def measured_traffic(groups, per_bucket):
    """HBM bytes per igemm launch from the committed PMC passes of this command (FETCH_SIZE x2 + WRITE_SIZE, separate passes over a
    serial un-graphed run with --groups 1 --per-bucket 40: tools/make_profiles.sh -> profiles/); None when the file is absent or
    the launch structure / batch differs from the profiled one."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            t = json.load(f)
        if groups != 1 or per_bucket != 40:
            return None
        c = t["hbm_bytes_corrected_per_dispatch"]
        return {"hbm_mb_per_launch": round(c["total"] / 1e6, 2), "read_mb": round(c["read"] / 1e6, 2), "write_mb": round(c["write"] / 1e6, 2),
                "source": f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, serial un-graphed run of this "
                          "workload with --groups 1; tools/make_profiles.sh)"}
    except Exception:   # noqa: BLE001
        return None


def bench_latency(est, ref, device):
    """Single-object latency (B = 1, 160x160, N = 1000) of the whole path with 2 and 4 refine iterations, captured as
    one hipGraph -- BASELINE configs[2] ("hipGraph-captured refine loop"); not part of `value`."""
    o = synth.make_object(4242, 160, 160, N_PTS, K_OBJ)
    d = {k: torch.from_numpy(o[k])[None].to(device) for k in ("img", "cloud", "choose", "obj")}
    out = {}
    for iters in (2, 4):
        pe = PoseEstimator(est, ref)
        res = (torch.empty(1, 7, dtype=torch.float64, device=device), torch.empty(1, 7, dtype=torch.float64, device=device))
        pe.estimate(d["img"], d["cloud"], d["choose"], d["obj"], iters, out=res)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pe.estimate(d["img"], d["cloud"], d["choose"], d["obj"], iters, out=res)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            pe.estimate(d["img"], d["cloud"], d["choose"], d["obj"], iters, out=res)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        out[f"iters{iters}_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
    out["shape"] = "B=1, 160x160 crop, N=1000, hipGraph replay, wall clock per pose"
    return out


def bench_knn():
    R, Q = 500, 500000
    knn = KNearestNeighbor(1)
    ref = (torch.rand(1, 3, R, device="cuda") - 0.5) * 0.2
    qry = (torch.rand(1, 3, Q, device="cuda") - 0.5) * 0.25
    for _ in range(3):
        knn(ref, qry)
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        knn(ref, qry)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byts, flops = 12 * (R + Q) + 8 * Q, 9.0 * R * Q
    gbs, tfl = byts / us / 1e3, flops / us / 1e6
    # BASELINE configs[4]: KNN stress, num_points = 2000 -> R = 500, Q = 1 000 000, 64 independent problems per GPU
    Bs, Qs = 64, 1000000
    refs = (torch.rand(Bs, 3, R, device="cuda") - 0.5) * 0.2
    qrys = (torch.rand(Bs, 3, Qs, device="cuda") - 0.5) * 0.25
    knn(refs, qrys)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        knn(refs, qrys)
    e1.record()
    torch.cuda.synchronize()
    ms_s = e0.elapsed_time(e1) / 3
    stress = {"batch": Bs, "R": R, "Q": Qs, "ms_per_launch": round(ms_s, 3), "Gpairs_per_s": round(Bs * R * Qs / ms_s / 1e6, 1),
              "achieved_GBps": round(Bs * (12 * (R + Qs) + 8 * Qs) / ms_s / 1e6, 1),
              "achieved_TFLOPs": round(9.0 * Bs * R * Qs / ms_s / 1e9, 2),
              "fp32_valu_frac": round(9.0 * Bs * R * Qs / ms_s / 1e9 / FP32_PEAK_TFLOPS, 4)}
    del refs, qrys
    # the clock the device delivers under a vector-ALU load right after the stress launches (the peak the fractions are priced
    # against assumes 2.4 GHz): box-to-box differences of the fraction show up here
    mhz = ctypes.c_double()
    _lib.check(_lib.lib().df_shader_clock_mhz(ctypes.byref(mhz), _lib.current_stream()), "shader_clock_mhz")
    stress["shader_clock_mhz_under_valu_load"] = round(mhz.value, 1)
    stress["fp32_valu_frac_at_delivered_clock"] = round(stress["fp32_valu_frac"] * 2400.0 / mhz.value, 4) if mhz.value > 0 else None
    return {"kernel": "knn1_dim3_sgpr_kernel (reference points through the scalar cache, 2 queries per lane)", "R": R, "Q": Q, "us_per_launch": round(us, 2), "stress_config5": stress,
            "algorithmic_bytes": byts, "achieved_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_flops": flops, "achieved_TFLOPs": round(tfl, 2), "fp32_valu_frac": round(tfl / FP32_PEAK_TFLOPS, 4),
            "bound": "fp32-valu (arithmetic intensity 9R/20 = 225 FLOP/B >> ridge ~20)"}


def bench_loss():
    """The symmetric PoseNet loss forward at BASELINE configs[3]'s size (N = 1000 per-point poses x M = 500 mesh points =
    250 M pairs; lib/loss.py:38-50 with the 1-NN of lib/knn): transform + 1-NN + distance reduction in one launch
    (csrc/loss.hip add_dis_sym_kernel, the scan of csrc/knn_core.h) + the small finishing kernel."""
    N, M = 1000, 500
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(7)
    q = torch.randn(N, 4, generator=g).to(dev); pt = (torch.randn(N, 3, generator=g) * 0.03).to(dev)
    pc = (torch.rand(N, generator=g) * 0.9 + 0.05).to(dev)
    mp = ((torch.rand(M, 3, generator=g) - 0.5) * 0.2).to(dev); tgt = mp + 0.5; pts = (torch.rand(N, 3, generator=g) * 0.1 + 0.45).to(dev)
    loss, dis = torch.empty(1, device=dev), torch.empty(1, device=dev)
    npts, ntgt, scratch = torch.empty(N, 3, device=dev), torch.empty(M, 3, device=dev), torch.empty(N, device=dev)
    sel = torch.empty(N, M, dtype=torch.int32, device=dev)
    L = _lib.lib()

    def call():
        _lib.check(L.df_loss_forward(q.data_ptr(), pt.data_ptr(), pc.data_ptr(), tgt.data_ptr(), mp.data_ptr(), pts.data_ptr(), N, M,
                                     ctypes.c_float(0.015), 1, loss.data_ptr(), dis.data_ptr(), npts.data_ptr(), ntgt.data_ptr(),
                                     scratch.data_ptr(), sel.data_ptr(), _lib.current_stream()), "loss_forward")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    n = 50
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    res = {"N": N, "M": M, "pairs": N * M * M, "us_per_forward": round(us, 2), "achieved_TFLOPs": round(9.0 * N * M * M / us / 1e6, 2),
           "fp32_valu_frac": round(9.0 * N * M * M / us / 1e6 / FP32_PEAK_TFLOPS, 4)}
    # what the training step runs since round 4: the symmetric frames of a pass stacked in ONE launch (df_loss_forward_frames; a pass of 8
    # frames has 4 symmetric ones in the bench's mix) -- the same per-frame arithmetic on a 4x larger grid
    F = 4
    rep = lambda t: t[None].repeat(F, *([1] * t.dim())).contiguous()
    qF, ptF, pcF, tgF, mpF, ptsF = rep(q), rep(pt), rep(pc), rep(tgt), rep(mp), rep(pts)
    lossF, disF = torch.empty(F, device=dev), torch.empty(F, device=dev)
    npF, ntF, scF = torch.empty(F, N, 3, device=dev), torch.empty(F, M, 3, device=dev), torch.empty(F, N, device=dev)
    selF = torch.empty(F, N, M, dtype=torch.int32, device=dev)
    symF = (ctypes.c_int * F)(*([1] * F))

    def call_frames():
        _lib.check(L.df_loss_forward_frames(F, symF, qF.data_ptr(), ptF.data_ptr(), pcF.data_ptr(), tgF.data_ptr(), mpF.data_ptr(), ptsF.data_ptr(), N, M,
                                            ctypes.c_float(0.015), lossF.data_ptr(), disF.data_ptr(), npF.data_ptr(), ntF.data_ptr(), scF.data_ptr(),
                                            selF.data_ptr(), _lib.current_stream()), "loss_forward_frames")
    for _ in range(3):
        call_frames()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        call_frames()
    e1.record()
    torch.cuda.synchronize()
    usF = e0.elapsed_time(e1) / 20 / F * 1e3
    res["stacked_4_frames"] = {"us_per_frame": round(usF, 2), "fp32_valu_frac": round(9.0 * N * M * M / usF / 1e6 / FP32_PEAK_TFLOPS, 4),
                               "same_numbers_as_single_frame": bool(torch.equal(disF[0], dis[0]) and torch.equal(lossF[F - 1], loss[0]))}
    return res


def bench_entry_point(est, ref, device, windows=9, frames=56):
    """The kept entry point's device pipeline (tools/eval_ycb.py -> densefusion_amd.lib.eval_window.WindowEstimator) on
    synthetic YCB-shaped frames that start in PINNED HOST memory: per window of `frames` keyframes (5 detections each, boxes
    cycling over the bench's seven crop sizes) one upload on a copy stream, device-side input preparation per crop-size
    bucket, ONE multi-bucket estimate call, one [n,7] download; the next window uploads while this one computes.  PNG decoding
    and .mat writing (host, disk) are outside -- the reference's own metric definition excludes disk I/O."""
    from densefusion_amd.lib.eval_window import WindowEstimator
    from densefusion_amd.lib.preprocess import get_bbox
    rng = np.random.Generator(np.random.PCG64(11))
    IH, IW = 480, 640
    rgb = torch.from_numpy(rng.integers(0, 256, (frames, IH, IW, 3), dtype=np.uint8)).pin_memory()
    depth = torch.from_numpy(rng.integers(5000, 12000, (frames, IH, IW)).astype(np.uint16).view(np.int16)).pin_memory()
    label_np = np.zeros((frames, IH, IW), dtype=np.int32)
    dets = []
    k = 0
    for f in range(frames):
        for j in range(5):
            H, W = CROPS[k % len(CROPS)]
            r0, c0 = int(rng.integers(1, IH - H + 1)), int(rng.integers(1, IW - W + 1))
            itemid = 1 + (k % K_OBJ)
            roi = np.array([0, itemid, c0, r0, c0 + W - 1, r0 + H - 1, 1.0])      # a PoseCNN row that get_bbox snaps to H x W
            bb = get_bbox(roi)
            assert (bb[1] - bb[0], bb[3] - bb[2]) == (H, W)
            box = label_np[f, bb[0]:bb[1], bb[2]:bb[3]]
            box[rng.random((H, W)) < 0.4] = itemid
            dets.append((f, itemid, roi, 1000 + k))
            k += 1
    label = torch.from_numpy(label_np).pin_memory()
    from collections import deque
    depth_ = int(os.environ.get("DF_BENCH_ENTRY_DEPTH", "4"))
    we = WindowEstimator(est, ref, N_PTS, ITERS, frames, (IH, IW), depth=depth_)
    for _ in range(depth_ + 1):                           # warm-up: every slot's workspace sized
        WindowEstimator.collect(we.submit(rgb, depth, label, dets))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending, lost = deque(), 0
    for _ in range(windows):
        while len(pending) >= depth_:
            lost += int(WindowEstimator.collect(pending.popleft())[2].sum())
        pending.append(we.submit(rgb, depth, label, dets))
    while pending:
        lost += int(WindowEstimator.collect(pending.popleft())[2].sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(dets) * windows
    return {"entry_point_poses_per_s": round((n - lost) / dt, 1), "frames_per_window": frames, "detections_per_window": len(dets),
            "windows": windows, "windows_in_flight": depth_, "lost": lost, "ms_per_window": round(dt / windows * 1e3, 2),
            "h2d_mb_per_window": round((rgb.numel() + depth.numel() * 2 + label.numel() * 4) / 1e6, 1),
            "note": "pinned host frames -> upload (copy stream) -> device input preparation -> one multi-bucket estimate -> poses to host; "
                    "what tools/eval_ycb.py runs per --window, without PNG decoding / .mat writing"}


def bench_train(device):
    """BASELINE configs[3] per GPU: YCB training step, K=21, N=1000, M=500 (PoseNet phase), symmetric KNN loss on half of the
    frames, 8 frames accumulated per optimizer step (tools/train.py:131-170).  The step is the NATIVE one (csrc/train.hip:
    forward + loss + backward of the frames of a pass in one library call, gradients accumulated in the flat kernel-layout
    buffer, Adam on that buffer): frames/s with the reference's bs = 1 passes (one call per frame: on one stream, and with the
    passes of a window on 4 lanes -- native_train.Lanes: own stream, host thread, workspace and gradient buffer each, gradients
    summed in lane order before the optimizer step) and with the 8 frames of a window sharing one pass.  `autograd_tape`: the round-2 path (every layer a
    Python autograd Function) on the same frames, with its per-kernel-kind TFLOP/s from HIP events around the conv launches."""
    from densefusion_amd import train_ops, train_utils
    from densefusion_amd.lib.loss import Loss
    from densefusion_amd.native_train import NativeTrainer
    K, N, M, acc = K_OBJ, N_PTS, 500, 8
    sym_list = [12, 15, 18, 19, 20]
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), WSEED).items()}
    H, W = 160, 160
    objs = []
    for j in range(acc):
        o = synth.make_object(500 + j, H, W, N, K, M)
        o["obj"][0] = [12, 3, 15, 7][j % 4]                         # every other frame is a symmetric object (KNN loss branch)
        objs.append(o)
    sym = [int(o["obj"][0]) in sym_list for o in objs]
    fr = {k: torch.stack([torch.from_numpy(o[k]) for o in objs]).to(device) for k in ("img", "cloud", "choose", "obj", "target", "model_points")}
    tr = NativeTrainer("posenet", N, K, device)
    tr.load_state_dict(sd)
    opt = train_utils.FlatAdam(tr, lr=1e-4)
    from densefusion_amd.native_train import Lanes
    nstreams = int(os.environ.get("DF_BENCH_TRAIN_STREAMS", "4"))
    lanes = Lanes(tr, nstreams)                  # pass j of a window on lane j % n: own stream, host thread, workspace, gradient buffer

    def frames(sl):
        return [fr[k][sl] for k in ("img", "cloud", "choose", "obj", "target", "model_points")]

    def window(P, multi=False):
        jobs = [(lambda lane, i=i: lane.step_posenet(*frames(slice(i, i + P)), sym[i:i + P], 0.015, dropout=True)) for i in range(0, acc, P)]
        if multi:
            lanes.run(jobs)
        else:
            for f in jobs:
                f(tr)
        train_utils.allreduce_gradients(tr); opt.step(grad_scale=1.0 / acc); tr.zero_grad()

    out = {"workload": f"YCB training step, K={K}, N={N}, M={M}, crop {H}x{W}, {acc} frames per optimizer step, fp32, native step (csrc/train.hip)",
           "frames_per_s": {}}
    for name, P, multi in (("1_per_pass", 1, True), ("1_per_pass_one_stream", 1, False), ("8_per_pass", acc, False)):
        window(P, multi)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            window(P, multi)
        torch.cuda.synchronize()
        out["frames_per_s"][name] = round(reps * acc / (time.perf_counter() - t0), 1)
    out["lanes_1_per_pass"] = nstreams
    # a window of frames of DIFFERENT crop sizes (what real data gives: no two frames can share a pass): bs = 1 passes on the lanes
    mixed = []
    for j in range(acc):
        Hm, Wm = CROPS[j % len(CROPS)]
        o = synth.make_object(700 + j, Hm, Wm, N, K, M)
        o["obj"][0] = [12, 3, 15, 7][j % 4]
        mixed.append(({k: torch.from_numpy(o[k])[None].to(device) for k in ("img", "cloud", "choose", "obj", "target", "model_points")},
                      int(o["obj"][0]) in sym_list))

    def mixed_window():
        lanes.run([(lambda lane, f=f, sy=sy: lane.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [sy], 0.015,
                                                              dropout=True)) for f, sy in mixed])
        train_utils.allreduce_gradients(tr); opt.step(grad_scale=1.0 / acc); tr.zero_grad()

    mixed_window(); mixed_window()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        mixed_window()
    torch.cuda.synchronize()
    out["frames_per_s"]["1_per_pass_mixed_crop_sizes_on_lanes"] = round(4 * acc / (time.perf_counter() - t0), 1)
    lanes.close()
    del lanes

    # the same mixed window as ONE multi-bucket pass (df_posenet_train_step_multi: what tools/train.py runs by default): bucketing and
    # stacking of the window's frames included
    mixed_frames = [dict(img=f["img"][0], cloud=f["cloud"][0], choose=f["choose"][0], obj=f["obj"][0], target=f["target"][0],
                         model_points=f["model_points"][0], symmetric=sy) for f, sy in mixed]

    def mixed_one_pass():
        tr.step_posenet_window(mixed_frames, 0.015, dropout=True)
        train_utils.allreduce_gradients(tr); opt.step(grad_scale=1.0 / acc); tr.zero_grad()

    mixed_one_pass(); mixed_one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        mixed_one_pass()
    torch.cuda.synchronize()
    out["frames_per_s"]["1_per_pass_mixed_crop_sizes"] = round(6 * acc / (time.perf_counter() - t0), 1)
    out["mixed_crop_sizes"] = [list(CROPS[j % len(CROPS)]) for j in range(acc)]

    # this fork's own default accumulation window (tools/train.py:34 of the reference: --batch_size 32): 32 frames of mixed crop sizes as one pass
    big = []
    for j in range(32):
        Hm, Wm = CROPS[j % len(CROPS)]
        o = synth.make_object(800 + j, Hm, Wm, N, K, M)
        o["obj"][0] = [12, 3, 15, 7][j % 4]
        big.append(dict(img=torch.from_numpy(o["img"]).to(device), cloud=torch.from_numpy(o["cloud"]).to(device), choose=torch.from_numpy(o["choose"]).to(device),
                        obj=torch.from_numpy(o["obj"]).to(device), target=torch.from_numpy(o["target"]).to(device),
                        model_points=torch.from_numpy(o["model_points"]).to(device), symmetric=int(o["obj"][0]) in sym_list))

    def window32():
        tr.step_posenet_window(big, 0.015, dropout=True)
        train_utils.allreduce_gradients(tr); opt.step(grad_scale=1.0 / 32); tr.zero_grad()

    window32(); window32()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        window32()
    torch.cuda.synchronize()
    out["frames_per_s"]["window_of_32_mixed_crop_sizes"] = round(3 * 32 / (time.perf_counter() - t0), 1)

    # refiner phase (tools/train.py:139-159 of the reference; batch_size / iteration = 16 frames per optimizer step at this fork's defaults): the
    # frozen estimator over the window's mixed crop sizes in one multi-bucket forward, Loss(refine=True) for all frames in one call, then `iteration` = 2 native
    # refiner steps over all 16 frames -- what tools/train.py --refine_start runs per window
    from densefusion_amd.lib.network import PoseNet as _PoseNet
    est = _PoseNet(N, K); est.load_state_dict(sd); est = est.to(device).eval()
    rtr = NativeTrainer("refiner", N, K, device)
    rtr.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K), WSEED + 1000).items()})
    ropt = train_utils.FlatAdam(rtr, lr=1e-4)
    crit = Loss(M, sym_list)
    rframes = big[:16]
    by_size = {}
    for f in rframes:
        by_size.setdefault(tuple(f["img"].shape[-2:]), []).append(f)
    rorder = [f for g in by_size.values() for f in g]
    robj = [int(f["obj"].reshape(-1)[0]) for f in rorder]          # host object indices (the loader's hint in tools/train.py: no read-back in the loop)

    def refine_window():
        stack = lambda k: torch.stack([f[k] for f in rorder])
        obj, mp = stack("obj"), stack("model_points")
        with torch.no_grad():
            pr, pt, pc, emb = est.forward_multi([torch.stack([f["img"] for f in g]) for g in by_size.values()], stack("cloud"), stack("choose"), obj)
            _, _, npts, ntg = crit.forward_frames(pr, pt, pc, stack("target"), mp, robj, stack("cloud"), 0.015, True)
        for _ in range(2):
            o = rtr.step_refiner(npts, emb, obj, ntg, mp, [f["symmetric"] for f in rorder])
            npts, ntg = o["new_points"], o["new_target"]
        train_utils.allreduce_gradients(rtr); ropt.step(grad_scale=1.0 / len(rorder)); rtr.zero_grad()

    refine_window(); refine_window()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        refine_window()
    torch.cuda.synchronize()
    out["frames_per_s"]["refiner_phase_window_of_16_mixed_crop_sizes"] = round(6 * len(rorder) / (time.perf_counter() - t0), 1)
    del est, rtr, ropt

    # roofline of the native step on EXECUTED FLOPs (df_trainer_profile: HIP events around every MFMA launch of the step on its stream;
    # FLOPs = 2 M N K of the shapes really launched -- low-resolution up-convolutions, folded head layer 1, chosen-pixel up_3,
    # F(4x4,3x3)-domain products -- not the reference graph's)
    def profiled(fn, reps=3):
        tr.profile(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        prof = tr.profile_read()
        tr.profile(False)
        return {k: (ms / reps, fl / reps, n // reps) for k, (ms, fl, n) in prof.items()}

    roof = {}
    for name, fn, fps_key, nfr in (("8_per_pass", lambda: window(acc, False), "8_per_pass", acc),
                                   ("mixed_crop_sizes_one_pass", mixed_one_pass, "1_per_pass_mixed_crop_sizes", acc),
                                   ("window_of_32_mixed_crop_sizes", window32, "window_of_32_mixed_crop_sizes", 32)):
        prof = profiled(fn)
        ms_all = sum(v[0] for v in prof.values()); fl_all = sum(v[1] for v in prof.values())
        wall_ms = nfr / out["frames_per_s"][fps_key] * 1e3
        roof[name] = {"kernel": "igemm_f32_v4 / v2 (forward, data gradient) + wgrad_f32_v2 (weight gradient), all MFMA launches of one optimizer window",
                      "bound": "mfma", "unit": "TFLOP/s", "peak": FP32_PEAK_TFLOPS,
                      "achieved": round(fl_all / ms_all / 1e9, 2), "frac": round(fl_all / ms_all / 1e9 / FP32_PEAK_TFLOPS, 4),
                      "executed_gflop_per_window": round(fl_all / 1e9, 1), "mfma_ms_per_window": round(ms_all, 3), "wall_ms_per_window": round(wall_ms, 3),
                      "whole_step_frac": round(fl_all / wall_ms / 1e9 / FP32_PEAK_TFLOPS, 4),
                      "per_kind": {k: {"ms": round(ms, 3), "launches": n, "tflops": round(fl / ms / 1e9, 2) if ms > 0 else 0.0,
                                       "frac": round(fl / ms / 1e9 / FP32_PEAK_TFLOPS, 4) if ms > 0 else 0.0} for k, (ms, fl, n) in prof.items()}}
    out["roofline"] = roof

    # the autograd-tape path of round 2 on the same frames (comparison + per-kernel-kind rates)
    net = PoseNet(N, K)
    net.load_state_dict(sd)
    net.to(device).train()
    flat = train_utils.FlatParams(net)
    opt2 = train_utils.FlatAdam(flat, lr=1e-4)
    crit = Loss(M, sym_list)
    fds = []
    for o in objs:
        fd = {k: torch.from_numpy(v).to(device) for k, v in o.items()}
        train_utils.with_host_index(fd["obj"], o["obj"])
        fds.append(fd)

    def tape_step(fs):
        with train_ops.splitk_scope(device):
            img = torch.stack([f["img"] for f in fs]); cloud = torch.stack([f["cloud"] for f in fs])
            choose = torch.stack([f["choose"] for f in fs]); obj = torch.stack([f["obj"] for f in fs])
            r, t, c, emb = net(img, cloud, choose, obj)
            loss = 0
            for b, f in enumerate(fs):
                loss = loss + crit(r[b:b + 1], t[b:b + 1], c[b:b + 1], f["target"][None], f["model_points"][None], f["obj"], f["cloud"][None],
                                   0.015, False)[0]
            loss.backward()

    def tape_window(P):
        for i in range(0, acc, P):
            tape_step(fds[i:i + P])
        opt2.step(); flat.zero_grad()

    tape = {}
    for P in (1, acc):
        tape_window(P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            tape_window(P)
        torch.cuda.synchronize()
        tape[f"{P}_per_pass"] = round(2 * acc / (time.perf_counter() - t0), 1)
    out["autograd_tape_frames_per_s"] = tape
    kinds, whole = {}, {}
    for P in (1, acc):
        train_ops.profile_begin()
        tape_window(P)
        prof = train_ops.profile_end()
        for kind, (ms, fl, n) in prof.items():
            kinds.setdefault(kind, {})[f"{P}_per_pass"] = {"ms_per_window": round(ms, 3), "launches": n, "tflops": round(fl / ms / 1e9, 2) if ms > 0 else 0.0,
                                                           "frac": round(fl / ms / 1e9 / FP32_PEAK_TFLOPS, 4) if ms > 0 else 0.0}
        # the whole NATIVE step against the matrix peak: FLOPs of the reference layer graph's conv forward / data gradient / weight
        # gradient launches (what the tape path executes; the native step executes fewer: low-resolution up-convs, folded head
        # layer 1, chosen-pixel up_3) / wall time of an optimizer window
        wall_s = acc / out["frames_per_s"][f"{P}_per_pass"]
        fl = sum(v[1] for v in prof.values())
        whole[f"{P}_per_pass"] = {"conv_gflop_per_frame_reference_graph": round(fl / acc / 1e9, 1), "tflops_wall": round(fl / wall_s / 1e12, 2),
                                  "reference_graph_flops_over_native_wall_frac": round(fl / wall_s / 1e12 / FP32_PEAK_TFLOPS, 4)}
    out["autograd_tape_conv_kernels"] = kinds
    out["whole_step_reference_graph"] = whole
    return out


def bench_config(K, N, crops, per_bucket, iters, device, wseed, cam=None, steps=6, inflight=2):
    """poses/s of the whole path (PoseNet -> selection -> `iters` refine passes, df_estimate_poses_multi) on another BASELINE
    configuration: `per_bucket` resident synthetic objects of every crop size in `crops` per step, the step captured as a hipGraph,
    `inflight` instances alternating like the headline loop.  A side measurement: a few steps, never `value`."""
    est, ref = PoseNet(N, K), PoseRefineNet(N, K)
    est.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), wseed).items()})
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K), wseed + 1000).items()})
    est, ref = est.to(device).eval(), ref.to(device).eval()
    imgs, cloud, choose, obj = [], [], [], []
    for ci, (H, W) in enumerate(crops):
        kw = {"cam": cam} if cam is not None else {}
        objs = [synth.make_object(9000 + 100 * ci + i, H, W, N, K, **kw) for i in range(per_bucket)]
        imgs.append(torch.from_numpy(np.stack([o["img"] for o in objs])).to(device))
        cloud += [o["cloud"] for o in objs]; choose += [o["choose"] for o in objs]; obj += [o["obj"] for o in objs]
    cloud, choose, obj = (torch.from_numpy(np.stack(a)).to(device) for a in (cloud, choose, obj))
    n = per_bucket * len(crops)
    insts = []
    for _ in range(inflight):
        pe = PoseEstimator(est, ref)
        res = (torch.empty(n, 7, dtype=torch.float64, device=device), torch.empty(n, 7, dtype=torch.float64, device=device))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            pe.estimate_multi(imgs, cloud, choose, obj, iters, out=res)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                pe.estimate_multi(imgs, cloud, choose, obj, iters, out=res)
        insts.append((pe, res, st, g))
    torch.cuda.synchronize()

    def run(k):
        for i in range(k):
            _, _, st, g = insts[i % inflight]
            with torch.cuda.stream(st):
                g.replay()
        torch.cuda.synchronize()

    run(inflight)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    finite = bool(torch.isfinite(insts[0][1][1]).all())
    return {"poses_per_s": round(n * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "objects_per_step": n, "num_obj": K, "num_points": N,
            "refine_iters": iters, "crops": [list(c) for c in crops], "steps": steps, "steps_in_flight": inflight, "hipgraph": True,
            "poses_finite": finite}


def bench_side_configs(device):
    """The other BASELINE.json configurations as cheap side objects (the headline stays configs[2]'s shape at 2 iterations)."""
    out = {}
    out["linemod_cfg2"] = dict(bench_config(13, 500, [(80, 80), (120, 120), (120, 160), (160, 160)], 40, 2, device, WSEED + 7, cam=synth.LINEMOD_CAM),
                               what="BASELINE configs[1]: LineMOD 13 objects, num_points=500, 2 refine iters, fp32, 4 crop sizes x 40 objects per step")
    out["ycb_cfg3_iters4"] = dict(bench_config(K_OBJ, N_PTS, CROPS, 40, 4, device, WSEED),
                                  what="BASELINE configs[2] as written: YCB 21 objects, num_points=1000, 4 refine iters, the headline's 7 crop sizes x 40 objects")
    out["n2000_b64_forward"] = dict(bench_config(K_OBJ, 2000, [(240, 320)], 64, 2, device, WSEED + 3, steps=4),
                                    what="BASELINE configs[4] shape through the forward path: num_points=2000, batch 64 objects of 240x320 per step, "
                                         "2 refine iters (its KNN stress is knn.stress_config5)")
    return out


def host_threads():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box
    exposes all host cores in os.cpu_count() but grants a 16-core share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:   # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("DF_CPU_THREADS", "16"))))


def cpu_baseline(buckets, gpu_poses, budget_s=float(os.environ.get("DF_BENCH_CPU_BUDGET_S", "14"))):
    """The CPU oracle (port of the reference path) on this host: one pose per crop size per round."""
    from oracle import dfnet, pose_math
    threads = host_threads()
    torch.set_num_threads(threads)
    sdp = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(K_OBJ), WSEED))
    sdr = dfnet._to_torch_sd(synth.make_state_dict(synth.refiner_spec(K_OBJ), WSEED + 1000))

    def one(b, i):
        h = b["host"]
        args = tuple(torch.from_numpy(h[k][i:i + 1]) for k in ("img", "cloud", "choose", "obj"))
        with torch.no_grad():
            return pose_math.estimate_pose(sdp, sdr, *args, ITERS)

    one(buckets[0], 0); one(buckets[3], 0)             # warm-up
    n, worst_add, checked = 0, 0.0, 0
    t0 = time.perf_counter()
    rnd = 0
    while True:
        for bi, b in enumerate(buckets):
            i = rnd % b["img"].shape[0]
            _, pose = one(b, i)
            n += 1
            mp = b["host"]["model_points"][i]
            add = pose_math.add_metric(pose_math.transform_model(pose, mp), pose_math.transform_model(gpu_poses[bi][i], mp))
            worst_add = max(worst_add, add); checked += 1
        rnd += 1
        if time.perf_counter() - t0 > budget_s or rnd >= 200:
            break
    dt = time.perf_counter() - t0
    extra = cpu_baseline_extras(threads)
    return ({"value": round(n / dt, 3), "unit": "poses/s", "cores": threads, "kind": "port", **extra,
             "sample": f"{n} poses = {rnd} round(s) of one pose per crop size {CROPS}, K=21, N=1000, 2 refine iters, "
                       f"oracle/ (torch CPU fp32) in {dt:.1f} s"},
            {"max_add_m_vs_oracle": float(f"{worst_add:.3e}"), "objects_checked": checked, "tolerance_m": 1e-4})


def cpu_baseline_extras(threads):
    """SURVEY 8d's other two CPU figures, a few seconds each: BASELINE configs[0] (LineMOD, K=13, N=500, 80x80, PoseNet
    forward only -- the reference's own CPU-runnable case) and the 1-NN at R=500, Q=500 000 (oracle/knn_ref.c, OpenMP)."""
    from oracle import dfnet
    from oracle.knn import knn_ref
    sd = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(13), WSEED + 7))
    o = synth.make_object(99, 80, 80, 500, 13, cam=synth.LINEMOD_CAM)
    args = tuple(torch.from_numpy(o[k])[None] if k in ("img", "cloud") else torch.from_numpy(o[k]) for k in ("img", "cloud", "choose", "obj"))
    with torch.no_grad():
        dfnet.posenet_forward(sd, *args)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.5:
            dfnet.posenet_forward(sd, *args)
            reps += 1
        cfg1 = reps / (time.perf_counter() - t0)
    rng = np.random.Generator(np.random.PCG64(5))
    ref = rng.random((1, 3, 500), dtype=np.float32)
    qry = rng.random((1, 3, 500000), dtype=np.float32)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    knn_ref(ref, qry[:, :, :1000], 1)
    t0 = time.perf_counter()
    knn_ref(ref, qry, 1)
    knn_ms = (time.perf_counter() - t0) * 1e3
    return {"config1_linemod_posenet_forward_per_s": round(cfg1, 2), "knn_500x500k_ms": round(knn_ms, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--per-bucket", type=int, default=40, help="objects of each crop size per step and GPU")
    ap.add_argument("--refine-iters", type=int, default=ITERS, help="refine iterations per pose (metric: 2; BASELINE configs[2] as written: 4)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("DF_BENCH_GROUPS", "1")),
                    help="split the step's objects into this many multi-bucket calls, one HIP stream each (0: one call per crop size, "
                         "the round-1 launch structure)")
    ap.add_argument("--no-streams", action="store_true", help="run the groups back to back on one stream")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("DF_BENCH_INFLIGHT", "4")),
                    help="steps in flight: consecutive steps (independent batches of the frame stream) alternate between this many "
                         "instances (own workspaces, output buffers, stream, hipGraph), so one step's memory-bound kernels overlap "
                         "the next step's GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank path with several ranks on one GPU)")
    args = ap.parse_args()

    globals()["ITERS"] = args.refine_iters
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: start it as `python bench.py --gpus N` (it starts its own "
                 "ranks) or under torch.distributed.run with --nproc-per-node N")
    if "DF_BENCH_DEVICE" in os.environ:            # rehearsal only: several ranks sharing one card
        local = int(os.environ["DF_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    # DF_BENCH_FORCE_DIST=1 (test switch): run the N > 1 code path -- process group, per-step gather, barriers -- at world size 1
    dist_on = world > 1 or bool(os.environ.get("DF_BENCH_FORCE_DIST"))
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":                  # "nccl" is RCCL on ROCm
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    est, ref = load_nets(device)
    buckets = make_buckets(rank, world, args.per_bucket, device)
    poses_per_step = args.per_bucket * len(CROPS)
    gdev = device if args.backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty(poses_per_step, 7, dtype=torch.float64, device=gdev) for _ in range(world)] if dist_on else None

    class Instance:
        """One step in flight: its own group workspaces, output buffers, stream and (captured) hipGraph; the inputs are shared."""

        def __init__(self):
            self.groups = make_groups(buckets, args.groups, device)
            self.pe = [PoseEstimator(est, ref) for _ in self.groups]          # one workspace per group (they may run concurrently)
            self.gstreams = None if (args.no_streams or len(self.groups) == 1) else [torch.cuda.Stream() for _ in self.groups]
            self.stream = inst_streams.pop() if inst_streams else torch.cuda.Stream()
            self.graph, self.pending = None, False
            run_step(self.pe, self.groups)                                     # eager pass: uploads weights, sizes the workspaces
            torch.cuda.synchronize()
            if not args.no_graph:
                # a capture failure is FATAL: the headline is defined on the captured step (config.hipgraph); `--no-graph` is the
                # only way to run it eagerly
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run_step(self.pe, self.groups, self.gstreams)
                torch.cuda.current_stream().wait_stream(side)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may touch the runtime
                    run_step(self.pe, self.groups, self.gstreams)

        def launch(self):
            """enqueue one step on this instance's stream (no host sync)"""
            with torch.cuda.stream(self.stream):
                if self.graph is not None:
                    self.graph.replay()
                else:
                    run_step(self.pe, self.groups, self.gstreams)

        def gather(self):      # results to every rank: the only communication of the inference path (a few KB per step)
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.stream)
            mine = torch.cat([g["out"][1] for g in self.groups])
            dist.all_gather(gathered, mine if args.backend == "nccl" else mine.cpu())
            self.stream.wait_stream(cur)          # the next step on this instance overwrites the outputs: after they were read
            self.pending = False

    from densefusion_amd.streams import concurrent_streams
    # the steps in flight overlap only if their streams sit on different hardware queues: tested, not assumed (streams.py)
    inst_streams = [] if os.environ.get("DF_BENCH_PLAIN_STREAMS") else concurrent_streams(device, max(1, args.inflight))
    insts = [Instance() for _ in range(max(1, args.inflight))]
    groups, pe = insts[0].groups, insts[0].pe
    graph, streams = insts[0].graph, insts[0].gstreams

    def timed_step(i):
        # N > 1: a step's results are gathered when its instance comes round again (or by drain() at the end): by then the step has
        # finished, so the wait the gather puts into the current stream -- which shares one of the runtime's 4 hardware queues with
        # an instance's stream -- holds nothing up.  Every step is gathered exactly once, inside the timed region.
        inst = insts[i % len(insts)]
        if dist_on and inst.pending:
            inst.gather()
        inst.launch()
        inst.pending = dist_on

    def drain():
        for inst in insts:
            if inst.pending:
                inst.gather()

    for i in range(args.warmup):
        timed_step(i)
    drain()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed_step(i)
    drain()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    witness = None
    if dist_on:
        my_dt = dt
        tmax = torch.tensor([dt], device=gdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # witness that the collective backend saw `world` DISTINCT devices: every rank contributes its device's UUID (16 bytes) and
        # its own timed-region rate through one all_gather on the backend under test
        props = torch.cuda.get_device_properties(device)
        import hashlib
        raw = hashlib.md5(f"{getattr(props, 'uuid', '')}|{getattr(props, 'pci_bus_id', '')}|{getattr(props, 'pci_device_id', '')}|"
                          f"{getattr(props, 'pci_domain_id', '')}|{props.name}".encode()).digest()
        mine = torch.tensor(list(raw[:16]) + [0.0], dtype=torch.float64, device=gdev)
        mine[16] = poses_per_step * args.steps / my_dt
        seen = [torch.empty(17, dtype=torch.float64, device=gdev) for _ in range(world)]
        dist.all_gather(seen, mine)
        ids = {bytes(int(v) for v in t[:16].tolist()).hex() for t in seen}
        rates = [float(t[16]) for t in seen]
        witness = {"ranks_seen": len(seen), "distinct_devices": len(ids), "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
                   "per_rank_poses_per_s": {"min": round(min(rates), 1), "max": round(max(rates), 1)},
                   "device_uuids": sorted(ids)}

    if rank == 0:
        total_poses = poses_per_step * world * args.steps
        gflop_step = sum(algorithmic_gflop_per_pose(H, W) for H, W in CROPS) * args.per_bucket
        out = {
            "metric": "poses/sec (node) YCB-Video 1000 pts + 2 refine iters",
            "value": round(total_poses / dt, 2), "unit": "poses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ycb_video_synthetic_stream (BASELINE configs[2] shape at the metric's 2 refine iters: "
                                   "K=21 objects, N=1000 points, crops cycled over 80x80..240x320, 5 objects/frame; every step "
                                   "re-evaluates the same objects_per_step_per_gpu resident objects)",
                       "num_obj": K_OBJ, "num_points": N_PTS, "refine_iters": ITERS, "crops": CROPS,
                       "objects_per_step_per_gpu": poses_per_step, "frames_per_step_per_gpu": poses_per_step / 5,
                       "hipgraph": graph is not None, "groups": len(groups), "group_streams": streams is not None, "steps_in_flight": len(insts), "sharding": f"objects round-robin over {world} rank(s), no data-path collective",
                       "reference_algorithm_gflop_per_step_per_gpu": round(gflop_step, 1),
                       "note": "reference_algorithm_* counts the FLOPs of the reference's own layer graph (SURVEY 8d); this build "
                               "executes fewer (PSP fold, low-resolution up-convs, Winograd F(4x4,3x3) / F(2x2,3x3) trunk, chosen-pixel up_3, confidence-first heads: DESIGN.md 5), so that rate may exceed the fp32 peak"},
            "reference_algorithm_tflops_per_gpu": round(gflop_step * args.steps / dt / 1e3, 2),
        }
        ms, fl, useful, by, n = profile_gemm(pe, groups, min(args.steps, 5))
        traffic = measured_traffic(len(groups), args.per_bucket)
        if traffic and n:
            traffic["ratio_to_algorithmic"] = round(traffic["hbm_mb_per_launch"] / (by / n / 1e6), 3)
        ach = fl / ms / 1e9 if ms > 0 else 0.0
        out["roofline"] = {"kernel": "igemm_f32_v4_kernel (implicit-GEMM conv / per-point / Winograd-domain GEMM, v_mfma_f32_32x32x2_f32; all launches of a step)",
                           "bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / FP32_PEAK_TFLOPS, 4),
                           "useful_frac": round(useful / ms / 1e9 / FP32_PEAK_TFLOPS, 4) if ms > 0 else 0.0,
                           # the whole step against the matrix peak: the FLOPs of the step's GEMM launches on rows that are not
                           # padding / the timed region's wall time per step (memory-bound kernels and launch gaps included)
                           "whole_step_frac": round(fl / max(1, min(args.steps, 5)) / (dt / args.steps) / 1e12 / FP32_PEAK_TFLOPS, 4),
                           "whole_step_useful_frac": round(useful / max(1, min(args.steps, 5)) / (dt / args.steps) / 1e12 / FP32_PEAK_TFLOPS, 4),
                           "traffic": traffic,
                           "algorithmic_mb_per_launch": round(by / max(n, 1) / 1e6, 2),
                           "launches_per_step": n // max(1, min(args.steps, 5)),
                           "avg_launch_us": round(ms / max(n, 1) * 1e3, 2),
                           "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 3),
                           "gemm_ms_per_step": round(ms / max(1, min(args.steps, 5)), 3)}
        assert out["n_gpus"] == args.gpus
        if witness is not None:
            out["collective_witness"] = witness
        if not args.no_knn and world == 1:          # the side measurements are single-GPU figures: not repeated per scaling point
            out["knn"] = bench_knn()
            out["knn"]["symmetric_loss_forward"] = bench_loss()
            out["latency_single_object"] = bench_latency(est, ref, device)
            out["entry_point"] = bench_entry_point(est, ref, device)
            out["train"] = bench_train(device)
            out["configs"] = bench_side_configs(device)
        if world == 1 and not args.no_cpu_baseline:
            gpu_poses = bucket_poses(buckets, groups)
            out["cpu_baseline"], out["parity"] = cpu_baseline(buckets, gpu_poses)
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
