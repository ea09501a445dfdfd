#!/bin/bash
# Dev tool (GPU box): per-kernel durations of the weight-gradient launches of chosen shapes (tools/dev/wgrad_shapes.py indices)
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_wg
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wg -- python3 tools/dev/wgrad_shapes.py "$@" > gpurun_out/prof_wg.log 2>&1
cut -c1-200 "$(find gpurun_out/prof_wg -name '*kernel_stats.csv' | head -1)" | head -8
rm -rf gpurun_out/prof_wg
cat gpurun_out/prof_wg.log | tail -3
