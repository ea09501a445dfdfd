#!/bin/bash
# Dev tool (runs ON the GPU box, from the repo root): the rocprofv3 evidence behind bench.py's roofline object.
#   1. kernel-trace + stats of the default bench command          -> gpurun_out/prof_bench/  (kernel_stats.csv)
#   2. FETCH_SIZE / WRITE_SIZE in separate --pmc passes over a serial, un-graphed run (the guide's HBM recipe)
#   3. condensed JSON summaries                                   -> gpurun_out/igemm_traffic.json
# Copy the summaries into profiles/ afterwards (named per round).
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_bench gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py > gpurun_out/prof_bench.log 2>&1
echo "[profiles] kernel trace done"
SER="--steps 2 --warmup 1 --no-cpu-baseline --no-knn --no-streams --no-graph --inflight 1 --groups 1"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $SER > gpurun_out/pmc_fetch.log 2>&1
echo "[profiles] FETCH_SIZE pass done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $SER > gpurun_out/pmc_write.log 2>&1
echo "[profiles] WRITE_SIZE pass done"
python3 tools/pmc_summary.py gpurun_out/igemm_traffic.json igemm_f32_v gpurun_out/pmc_fetch gpurun_out/pmc_write | cut -c1-400
cp "$(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1)" gpurun_out/bench_kernel_stats.csv
tail -1 gpurun_out/prof_bench.log | cut -c1-300
