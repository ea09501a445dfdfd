set -e
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py tests/test_conv_grad_gpu.py -q 2>&1 | tail -2
for t in a b c; do DF_IGEMM_LOWOCC=0 DF_IGEMM_TILE=$t timeout -k 10 400 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py -q 2>&1 | tail -1; done
bash tools/dev/gemm_list.sh r11 --groups 1 --inflight 1 | head -1
bash tools/dev/gemm_list.sh r11b --groups 1 --inflight 1 | head -1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o /tmp/igemm_trace tools/dev/igemm_trace.hip densefusion_amd/csrc/common.hip 2>/dev/null
for s in "131072 1024 1024" "286720 512 256" "286720 512 192"; do
  timeout -k 10 60 /tmp/igemm_trace $s | head -3 | grep -v "cycle counter"
done
