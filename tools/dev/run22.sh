set -eo pipefail
bash tools/make_profiles.sh
bash tools/dev/prof_serial.sh r02f --inflight 1 --groups 1 > /dev/null
python3 tools/glue_roofline.py gpurun_out/r02f_serial_kernel_stats.csv 12 gpurun_out/r02f_glue.json | tail -3
bash tools/dev/prof_train.sh r02f | tail -3
timeout -k 10 400 python bench.py > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
cat gpurun_out/r02f_bench.json | cut -c1-600
