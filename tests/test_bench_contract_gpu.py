"""GPU: bench.py prints ONE JSON line with the fields the driver reads (metric / value / roofline / cpu_baseline ...)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    env = dict(os.environ, DF_BENCH_CPU_BUDGET_S="3")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--per-bucket", "8"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "poses/s" and d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] <= 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "poses/s" and c["sample"]
    assert d["parity"]["max_add_m_vs_oracle"] < d["parity"]["tolerance_m"]
