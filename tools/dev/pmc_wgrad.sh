#!/bin/bash
# Dev tool (GPU box): counters of the weight-gradient kernel on one shape of tools/dev/wgrad_shapes.py (index): fabric-side bytes, L2 hit
# rate, SQ busy / wait / MFMA counters (separate --pmc passes)
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf gpurun_out/pw
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pw -- python3 tools/dev/wgrad_shapes.py "$@" > gpurun_out/pw.log 2>&1 || { tail -3 gpurun_out/pw.log; continue; }
  python3 - "$C" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/pw/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = {}
for r in rows:
    if 'wgrad_f32_v2' in r['Kernel_Name']:
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, 'per dispatch', sum(v) / len(v), '(n=%d)' % len(v))
PY
done
rm -rf gpurun_out/pw
