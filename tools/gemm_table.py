"""Dev tool: every GEMM launch of one bench step (all crop sizes, B objects each), sorted by time, with TFLOP/s."""
import os, sys, re, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import bench
    rows = []
    for (H, W) in bench.CROPS:
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "layer_profile.py"), str(H), str(W), str(B)],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        for ln in out.splitlines():
            m = re.match(r"\[df-gemm\] (M=\d+ N=\d+ K=\d+ k\dx\d s\d d\d z\d+)\s+([\d.]+) us\s+([\d.]+) TFLOP/s", ln)
            if m:
                rows.append((f"{H}x{W}", m.group(1), float(m.group(2)), float(m.group(3))))
    tot = sum(r[2] for r in rows)
    fl = sum(r[2] * r[3] for r in rows)
    print(f"{len(rows)} launches, {tot/1e3:.2f} ms, {fl/tot:.1f} TFLOP/s aggregate")
    acc = 0.0
    for r in sorted(rows, key=lambda r: -r[2])[:60]:
        acc += r[2]
        print(f"{r[0]:>8} {r[1]:<44} {r[2]:8.1f} us {r[3]:6.1f} TF  cum {100*acc/tot:5.1f}%")
    # time lost vs 125 TFLOP/s, by launch
    print("--- time above a 125 TFLOP/s pace, top 25")
    loss = sorted(rows, key=lambda r: -(r[2] - r[2] * r[3] / 125.0))[:25]
    for r in loss:
        print(f"{r[0]:>8} {r[1]:<44} {r[2]:8.1f} us {r[3]:6.1f} TF  excess {r[2] - r[2]*r[3]/125.0:7.1f} us")
    print(f"total excess {sum(r[2] - r[2]*r[3]/125.0 for r in rows)/1e3:.2f} ms of {tot/1e3:.2f} ms")
    print("--- by shape class (N, K, kernel, z): launches, time, TFLOP/s, excess")
    cls = {}
    for r in rows:
        key = " ".join(r[1].split()[1:])
        c = cls.setdefault(key, [0, 0.0, 0.0])
        c[0] += 1; c[1] += r[2]; c[2] += r[2] * r[3]
    for key, c in sorted(cls.items(), key=lambda kv: -(kv[1][1] - kv[1][2] / 125.0))[:24]:
        print(f"{key:<36} {c[0]:4d} {c[1]:9.1f} us {c[2]/c[1]:6.1f} TF  excess {c[1] - c[2]/125.0:8.1f} us")

if __name__ == "__main__":
    main()
