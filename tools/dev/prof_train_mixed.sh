#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel trace of the training step on windows of 8 mixed-size frames as one multi-bucket pass
# (tools/train_bench.py mixed) -> gpurun_out/<tag>_train_mixed_kernel_stats.csv
set -eo pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_train
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_bench.py mixed 4 > gpurun_out/prof_train.log 2>&1
cp "$(find gpurun_out/prof_train -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_train_mixed_kernel_stats.csv
rm -rf gpurun_out/prof_train
tail -1 gpurun_out/prof_train.log
cut -c1-160 gpurun_out/${TAG}_train_mixed_kernel_stats.csv | head -40
