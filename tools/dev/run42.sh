set -e
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py tests/test_conv_grad_gpu.py -q 2>&1 | tail -2
for t in a b c; do DF_IGEMM_LOWOCC=0 DF_IGEMM_TILE=$t timeout -k 10 400 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py -q 2>&1 | tail -1; done
bash tools/dev/gemm_list.sh r13 --groups 1 --inflight 1 | head -1
bash tools/dev/gemm_list.sh r13b --groups 1 --inflight 1 | head -1
