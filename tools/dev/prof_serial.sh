#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel trace of a SERIAL (one stream, no graph) bench run -> gpurun_out/<tag>_serial_kernel_stats.csv
set -eo pipefail
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_serial
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-knn --no-streams --no-graph "$@" > gpurun_out/prof_serial.log 2>&1
cp "$(find gpurun_out/prof_serial -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_serial_kernel_stats.csv
# steady state (last pass) per kernel; 6 + DF_BENCH_INFLIGHT eager passes + 5 profiled ones with the default flags above
python3 tools/trace_summary.py "$(find gpurun_out/prof_serial -name '*kernel_trace.csv' | head -1)" ${PASSES:-15} gpurun_out/${TAG}_serial_last_pass.json > gpurun_out/${TAG}_serial_last_pass.txt
rm -rf gpurun_out/prof_serial
tail -1 gpurun_out/prof_serial.log | cut -c1-400
cut -c1-150 gpurun_out/${TAG}_serial_kernel_stats.csv | head -30
