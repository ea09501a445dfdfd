set -eo pipefail
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_full_gpu.log 2>&1 || { tail -30 gpurun_out/r3_full_gpu.log; exit 1; }
tail -2 gpurun_out/r3_full_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/make_profiles.sh
bash tools/dev/prof_serial.sh r03 --inflight 1 --groups 1 > /dev/null
python3 tools/glue_roofline.py gpurun_out/r03_serial_kernel_stats.csv 12 gpurun_out/r03_glue.json | tail -1
bash tools/dev/prof_train.sh r03 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cut -c1-300 gpurun_out/r03_bench.json
