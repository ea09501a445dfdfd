set -e
timeout -k 10 400 python -m pytest tests/test_conv_gpu.py tests/test_network_gpu.py tests/test_conv_grad_gpu.py -x -q 2>&1 | tail -2
for t in a b c; do DF_IGEMM_LOWOCC=0 DF_IGEMM_TILE=$t timeout -k 10 400 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py -x -q 2>&1 | tail -1; done
bash tools/dev/gemm_list.sh r2x --groups 1 --inflight 1 > /dev/null
python - <<PY
import re,collections
def load(tag):
    rows=[]
    for ln in open(f"gpurun_out/{tag}_gemm_raw.txt"):
        m=re.match(r"\[df-gemm\] (M=\d+ N=\d+ K=\d+ k\dx\d s\d d\d z\d+)\s+([\d.]+) us", ln)
        if m: rows.append((m.group(1), float(m.group(2))))
    return rows
a=load("r2x")
print(len(a), sum(t for _,t in a))
agg=collections.OrderedDict()
for d,t in a:
    if "k1x1" in d: continue
    e=agg.setdefault(d,[0,0]); e[0]+=t; e[1]+=1
for d,e in agg.items(): print(d, e[1], round(e[0],1))
PY
