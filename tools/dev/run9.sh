set -e
timeout -k 10 1000 python -m pytest tests/test_rccl_gpu.py tests/test_train_tool_gpu.py -x -q > gpurun_out/r2_t9.log 2>&1 || { tail -60 gpurun_out/r2_t9.log; exit 1; }
tail -3 gpurun_out/r2_t9.log
