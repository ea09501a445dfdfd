for i in 1 2; do echo "$(timeout -k 10 200 python tools/train_bench.py 1 2>&1 | tail -1 | cut -c1-40) | $(timeout -k 10 200 python tools/train_bench.py 8 2>&1 | tail -1 | cut -c1-40)"; done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-knn 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['train']['frames_per_s'])"
