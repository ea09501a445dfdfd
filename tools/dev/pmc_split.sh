#!/bin/bash
# Dev tool (GPU box): matrix-pipe and LDS counters of the split-precision GEMM experiment on the probe's largest shape -> gpurun_out/split_pmc.txt
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DF_DEV_LIB=1 DF_GEMM_SPLIT_BF16=1
: > gpurun_out/split_pmc.txt
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  rm -rf gpurun_out/pmc_split
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_split -- python3 tools/dev/split_gemm_probe.py 0 > gpurun_out/pmc_split.log 2>&1 || true
  python3 - "$C" <<'PY' >> gpurun_out/split_pmc.txt
import csv, glob, sys, collections
f = glob.glob("gpurun_out/pmc_split/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file for", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if "gemm_split" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()):
    print(f"{kn:42s} {cn:28s} launches {n:3d} per-launch {v / n:.4g}")
PY
done
rm -rf gpurun_out/pmc_split
cat gpurun_out/split_pmc.txt
