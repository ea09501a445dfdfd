#!/bin/bash
# Dev tool (GPU box): per-launch GEMM table of one serial bench step -> gpurun_out/<tag>_gemm_list.txt
TAG=${1:-r02}; shift || true
DF_DEV_LIB=1 DF_PROFILE_VERBOSE=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-knn --no-graph "$@" 2> gpurun_out/${TAG}_gemm_raw.txt > gpurun_out/${TAG}_gemm_bench.json
python3 - <<PY
import re,collections
rows=[]
for ln in open("gpurun_out/${TAG}_gemm_raw.txt"):
    m=re.match(r"\[df-gemm\] (M=(\d+) N=(\d+) K=(\d+) k\dx\d s\d d\d z(\d+))\s+([\d.]+) us\s+([\d.]+) TFLOP/s", ln)
    if m: rows.append((m.group(1), float(m.group(6)), float(m.group(7))))
tot=sum(r[1] for r in rows); fl=sum(r[1]*r[2] for r in rows)
out=[f"{len(rows)} launches, {tot/1e3:.2f} ms, {fl/tot:.1f} TFLOP/s aggregate"]
acc=0
for r in sorted(rows,key=lambda r:-r[1]):
    acc+=r[1]
    out.append(f"{r[0]:<46} {r[1]:9.1f} us {r[2]:6.1f} TF  cum {100*acc/tot:5.1f}%  excess@130 {r[1]-r[1]*r[2]/130:8.1f}")
open("gpurun_out/${TAG}_gemm_list.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[:70]))
PY
