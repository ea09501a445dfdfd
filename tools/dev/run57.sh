timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_trainops_gpu.py tests/test_train_tool_gpu.py tests/test_loss_gpu.py -q 2>&1 | tail -3
for i in 1 2; do echo "$(timeout -k 10 200 python tools/train_bench.py 1 2>&1 | tail -1 | cut -c11-28) | $(timeout -k 10 200 python tools/train_bench.py 8 2>&1 | tail -1 | cut -c11-28)"; done
