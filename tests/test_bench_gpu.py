"""GPU: bench.py end to end on a reduced batch -- the driver's contract for the JSON line (keys, units, the two extra objects)
and the internal consistency of the numbers it reports."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    env = dict(os.environ, DF_BENCH_CPU_BUDGET_S="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--per-bucket", "4"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "poses/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value is the whole job's throughput: poses of the timed steps / their wall time
    poses = d["config"]["objects_per_step_per_gpu"] * d["steps"]
    assert abs(d["value"] - poses / (d["ms_per_step"] * d["steps"] / 1e3)) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and 0 < r["useful_frac"] <= r["frac"]
    assert 0 < r["whole_step_useful_frac"] <= r["whole_step_frac"] < 1     # (wall time with 4 steps in flight; at this tiny batch it can beat the serial GEMM sum)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_us"] * 1e3) <= 0.02 * r["achieved"]     # GFLOP / us = PFLOP/s
    assert r["traffic"] is None                       # the committed PMC profile is for the default batch only
    c = d["cpu_baseline"]
    assert c["unit"] == "poses/s" and c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["parity"]["max_add_m_vs_oracle"] < d["parity"]["tolerance_m"] == 1e-4
    for k in ("knn", "train", "entry_point", "latency_single_object"):
        assert k in d, k
    assert d["entry_point"]["lost"] == 0 and d["entry_point"]["entry_point_poses_per_s"] > 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes itself (gloo, both on the one card of a
    test box -- a rehearsal of the N-GPU command the driver issues) and relays rank 0's line with n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DF_BENCH_DEVICE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                          "--per-bucket", "2", "--inflight", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    poses = d["config"]["objects_per_step_per_gpu"] * 2 * d["steps"]                  # whole-job aggregate over both ranks
    assert abs(d["value"] - poses / (d["ms_per_step"] * d["steps"] / 1e3)) <= 0.01 * d["value"]
    assert "roofline" in d and "cpu_baseline" not in d                                 # the CPU leg runs at N = 1 only


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr and not out.stdout.strip()


def test_bench_multi_rank_code_path_on_rccl_at_world_size_one():
    """The N > 1 code path of bench.py -- process group on RCCL, the per-step all_gather of the poses next to four graph replays in
    flight (gathered when an instance comes round again, drained at the end), barriers, the max-over-ranks all_reduce -- executed at
    world size 1, the only RCCL world a one-GPU box offers (DF_BENCH_FORCE_DIST=1)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "9", "--warmup", "2", "--per-bucket", "4",
                          "--no-knn", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["steps"] == 9 and d["value"] > 0
