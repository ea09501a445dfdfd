#!/bin/bash
# Dev tool (GPU box): the launch sequence of ONE native training step (8 frames of 160x160) with grid sizes and durations
# -> gpurun_out/train_seq.txt
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DF_TB_ONE_SIZE=1
rm -rf gpurun_out/prof_seq
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_seq -- python3 tools/train_bench.py ${P:-8} 1 > gpurun_out/prof_seq.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_seq/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last nchw3_to_nhwc4 launch on
idx = [i for i, r in enumerate(rows) if "nchw3_to_nhwc4" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
with open("gpurun_out/train_seq.txt", "w") as o:
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        n = r["Kernel_Name"].replace("df::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        o.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  grid {r["Grid_Size_X"]:>8s} x{r["Grid_Size_Y"]} x{r["Grid_Size_Z"]}  {n}\n')
PY
rm -rf gpurun_out/prof_seq
wc -l gpurun_out/train_seq.txt
