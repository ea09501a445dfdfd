"""Dev tool: steady-state per-kernel time of one bench step from a rocprofv3 kernel trace.

    python tools/trace_summary.py KERNEL_TRACE.csv PASSES [OUT.json]

KERNEL_TRACE.csv: the *_kernel_trace.csv of `rocprofv3 --kernel-trace` over a SERIAL bench run (tools/dev/prof_serial.sh);
PASSES: passes over the step's objects in that run.  `--stats` averages include the first pass, whose launches touch fresh
workspace pages (single dispatches of several ms); this tool cuts every kernel's dispatches into PASSES equal runs in launch
order and reports the LAST one (calls, total us, and the largest single dispatch)."""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?(?:df::)?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    path, passes = sys.argv[1], int(sys.argv[2])
    csv.field_size_limit(1 << 30)
    per = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            per.setdefault(short(r["Kernel_Name"]), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows = []
    for k, v in per.items():
        v.sort()
        if len(v) % passes or len(v) < passes:
            continue                       # load-time kernels (weight packing) and torch's own
        n = len(v) // passes
        last = [d for _, d in v[-n:]]
        rows.append({"kernel": k, "calls_per_step": n, "us_per_step": round(sum(last) / 1e3, 1), "max_us": round(max(last) / 1e3, 1)})
    rows.sort(key=lambda r: -r["us_per_step"])
    tot = sum(r["us_per_step"] for r in rows)
    for r in rows:
        print(f"{r['kernel'][:70]:70s} {r['calls_per_step']:4d} {r['us_per_step']:9.1f} us  max {r['max_us']:8.1f}")
    print(f"total {tot:.1f} us per step")
    if len(sys.argv) > 3:
        json.dump({"source": path, "passes": passes, "total_us_per_step": round(tot, 1), "kernels": rows}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
