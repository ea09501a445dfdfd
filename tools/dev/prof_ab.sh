#!/bin/bash
# Dev tool (GPU box): kernel stats of tools/train_bench.py 8 (one crop size) with and without an environment switch.  usage: prof_ab.sh VAR
set -eo pipefail
VAR=${1:-DF_TRAIN_WGRAD_DIRECT}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DF_TB_ONE_SIZE=1
for mode in off on; do
  rm -rf gpurun_out/prof_ab
  if [ $mode = on ]; then export $VAR=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 tools/train_bench.py 8 4 > gpurun_out/prof_ab_$mode.log 2>&1
  cp "$(find gpurun_out/prof_ab -name '*kernel_stats.csv' | head -1)" gpurun_out/ab_${mode}_kernel_stats.csv
  rm -rf gpurun_out/prof_ab
  tail -1 gpurun_out/prof_ab_$mode.log
done
