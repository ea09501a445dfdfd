for PB in 40 60 80; do
  timeout -k 10 300 python bench.py --per-bucket $PB --no-cpu-baseline --no-knn --steps 12 > gpurun_out/r2l_pb$PB.json 2> gpurun_out/r2l_pb$PB.err || { tail -5 gpurun_out/r2l_pb$PB.err; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r2l_pb$PB.json"))
r=d["roofline"]
print("pb=$PB", d["value"], d["ms_per_step"], r["frac"], r["useful_frac"], r["gemm_ms_per_step"])
PY
done
