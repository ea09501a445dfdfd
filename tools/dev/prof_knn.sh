#!/bin/bash
# Dev tool (GPU box): counter evidence for the 1-NN kernels -> gpurun_out/<tag>_knn_pmc.json
#   three workloads (500 x 500 000 one launch; the 64 x 500 x 1 000 000 stress; the symmetric loss forward), each: a kernel trace
#   (durations) and three separate --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE), condensed by tools/pmc_summary.py
set -eo pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run() {   # name, kernel substring, program args...
  local name=$1 needle=$2; shift 2
  rm -rf gpurun_out/pk_${name}_*
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk_${name}_trace -- python3 "$@" > gpurun_out/pk_${name}.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pk_${name}_sq -- python3 "$@" >> gpurun_out/pk_${name}.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pk_${name}_fetch -- python3 "$@" >> gpurun_out/pk_${name}.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pk_${name}_write -- python3 "$@" >> gpurun_out/pk_${name}.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/${TAG}_knn_pmc_${name}.json "$needle" gpurun_out/pk_${name}_sq gpurun_out/pk_${name}_fetch gpurun_out/pk_${name}_write | cut -c1-300
  grep "$needle" "$(find gpurun_out/pk_${name}_trace -name '*kernel_stats.csv' | head -1)" | cut -c1-200 > gpurun_out/${TAG}_knn_stats_${name}.csv || true
  grep -a "us" gpurun_out/pk_${name}.log | head -2
  rm -rf gpurun_out/pk_${name}_*
}
run single knn1_dim3 tools/knn_microbench.py 1 500 500000
run stress knn1_dim3 tools/knn_microbench.py 64 500 1000000
run symloss add_dis_sym tools/loss_microbench.py
