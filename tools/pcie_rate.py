"""Dev tool: poses/s when the per-object inputs start in (pinned) host memory and results are copied back --
the PCIe-inclusive rate DESIGN.md quotes next to bench.py's HBM-resident `value` (never used as `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    dev = torch.device("cuda", 0)
    est, ref = bench.load_nets(dev)
    buckets = bench.make_buckets(0, 1, 10, dev)
    pe = [bench.PoseEstimator(est, ref) for _ in buckets]
    streams = [torch.cuda.Stream() for _ in buckets]
    host = [{k: torch.from_numpy(b["host"][k]).pin_memory() for k in ("img", "cloud", "choose", "obj")} for b in buckets]
    out_h = [torch.empty(10, 7, dtype=torch.float64).pin_memory() for _ in buckets]
    def step():
        main_s = torch.cuda.current_stream()
        for i in reversed(range(len(buckets))):
            st = streams[i]; st.wait_stream(main_s)
            with torch.cuda.stream(st):
                d = {k: v.to(dev, non_blocking=True) for k, v in host[i].items()}
                _, pose = pe[i].estimate(d["img"], d["cloud"], d["choose"], d["obj"], bench.ITERS)
                out_h[i].copy_(pose, non_blocking=True)
        for st in streams: main_s.wait_stream(st)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mb = sum(sum(v.numel() * v.element_size() for v in h.values()) for h in host) / 1e6
    print(f"PCIe-inclusive: {70*n/dt:.1f} poses/s ({dt/n*1e3:.2f} ms/step, {mb:.1f} MB host->device per step, eager multi-stream)")

if __name__ == "__main__":
    main()
