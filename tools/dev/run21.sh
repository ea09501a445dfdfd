set -e
timeout -k 10 600 python -m pytest tests/test_network_gpu.py -x -q 2>&1 | tail -3
bash tools/dev/prof_serial.sh r02e --inflight 1 --groups 1 | tail -22 | cut -c1-150
