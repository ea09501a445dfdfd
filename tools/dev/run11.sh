set -e
for cfg in "1 3" "1 4" "1 6" "1 8"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --groups $1 --inflight $2 --no-cpu-baseline --no-knn --steps 24 > gpurun_out/r2j_bench_g$1_f$2.json 2> gpurun_out/r2j_bench_g$1_f$2.err || { tail -5 gpurun_out/r2j_bench_g$1_f$2.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r2j_bench_g$1_f$2.json"))
r=d["roofline"]
print("G=$1 inflight=$2", d["value"], d["ms_per_step"], r["frac"], r["useful_frac"], r["launches_per_step"], r["gemm_ms_per_step"])
PY
done
