set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SER="--steps 2 --warmup 1 --no-cpu-baseline --no-knn --no-streams --no-graph --inflight 1 --groups 1"
rm -rf gpurun_out/pmc_f
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 bench.py $SER > gpurun_out/pmc_f.log 2>&1
rm -rf gpurun_out/pmc_w
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python3 bench.py $SER > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_summary.py gpurun_out/traffic_nt.json igemm_f32_v4_kernel gpurun_out/pmc_f gpurun_out/pmc_w | cut -c1-420
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
bash tools/dev/gemm_list.sh r2v --groups 1 --inflight 1 | head -12
