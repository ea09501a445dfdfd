#!/bin/bash
# Dev tool (GPU box): sample the GPU's shader clock and power while the bench runs -> gpurun_out/clock_watch.txt
cd "$GRAFT_REPO_ROOT"
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | sed 's/GPU\[0\]\s*: //' | tr '\n' '|'; echo; sleep 0.5; done ) > gpurun_out/clock_watch.txt &
W=$!
timeout -k 10 300 python bench.py --no-cpu-baseline --no-knn --steps ${STEPS:-200} "$@" > gpurun_out/clock_bench.json 2>/dev/null
kill $W 2>/dev/null
cat gpurun_out/clock_watch.txt | head -24
