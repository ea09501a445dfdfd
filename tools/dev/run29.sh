set -e
for n in 2 3 4 5; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-knn --inflight $n > gpurun_out/r5_$n.json 2> gpurun_out/r5.err || { tail -5 gpurun_out/r5.err; exit 1; }
python - <<PY
import json
j=json.load(open("gpurun_out/r5_$n.json"))
print("inflight $n", j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
done
