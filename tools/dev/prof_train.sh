#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel trace of the training step (tools/train_bench.py, 8 frames of 160x160 per pass: bench.py's training workload) -> gpurun_out/<tag>_train_kernel_stats.csv
set -eo pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DF_TB_ONE_SIZE=1
rm -rf gpurun_out/prof_train
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_bench.py 8 4 > gpurun_out/prof_train.log 2>&1
cp "$(find gpurun_out/prof_train -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_train_kernel_stats.csv
rm -rf gpurun_out/prof_train
tail -1 gpurun_out/prof_train.log
cut -c1-160 gpurun_out/${TAG}_train_kernel_stats.csv | head -24
