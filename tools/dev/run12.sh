set -e
timeout -k 10 600 python -m pytest tests/test_eval_ycb_tool_gpu.py tests/test_rccl_gpu.py -x -q > gpurun_out/r2_t12.log 2>&1 || { tail -40 gpurun_out/r2_t12.log; exit 1; }
tail -2 gpurun_out/r2_t12.log
timeout -k 10 400 python bench.py --steps 20 > gpurun_out/r2k_bench.json 2> gpurun_out/r2k_bench.err || { tail -20 gpurun_out/r2k_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r2k_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["useful_frac"], json.dumps(d["entry_point"])[:300], d["cpu_baseline"]["value"], d["parity"])
PY
