"""Dev tool: achieved HBM GB/s of the memory-bound glue kernels of one bench step against their algorithmic bytes.

    python tools/glue_roofline.py SERIAL_LAST_PASS.json OUT.json

SERIAL_LAST_PASS.json: tools/trace_summary.py over the kernel trace of a SERIAL bench run (tools/dev/prof_serial.sh: one stream, no
graph, --groups 1): per kernel the time of the run's last pass over the step's 280 objects (the first passes touch fresh pages).
Algorithmic bytes (every tensor a kernel must read / write, once, fp32) are computed here from the bench workload: K = 21,
N = 1000 (Npad = 1024), 40 objects of each of the seven crop sizes, 2 refine iterations."""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402

HBM = bench.HBM_PEAK_GBS
N, NPAD, PER, ITERS = bench.N_PTS, 1024, 40, bench.ITERS


def wino_geom(H, W, d, m):
    TH, TW = ((H + d - 1) // d + m - 1) // m, ((W + d - 1) // d + m - 1) // m
    return d * d * TH * TW


def wino_route(H, W, d, cin, cout):
    """the engine's own decision (csrc/wino.hip wino_route): 0 direct, 2 = F(2x2,3x3), 4 = F(4x4,3x3)"""
    from densefusion_amd import _lib
    return _lib.lib().df_wino_route(H, W, d, cin, cout)


def step_bytes():
    b = {}
    add = lambda k, v: b.__setitem__(k, b.get(k, 0.0) + v)
    for (H, W) in bench.CROPS:
        B = PER
        h2, w2 = H // 2, W // 2            # stem
        h4, w4 = H // 4, W // 4            # pool / layer1
        h, w = H // 8, W // 8              # layer2..4, psp
        add("nchw3_to_nhwc4_kernel", B * (3 + 4) * H * W * 4)
        add("maxpool3s2_kernel", B * (h2 * w2 + h4 * w4) * 64 * 4)
        # stride-1 3x3 convs that may take a Winograd route (engine.hip cnn_forward): (cin, cout, dil, residual)
        for cin, cout, d, res in ((128, 128, 1, True), (128, 128, 1, False), (128, 128, 1, True), (128, 256, 1, False), (256, 256, 1, True),
                                  (256, 256, 2, False), (256, 256, 2, True), (256, 512, 1, False), (512, 512, 1, True),
                                  (512, 512, 4, False), (512, 512, 4, True)):
            m = wino_route(h, w, d, cin, cout)
            if not m:
                continue
            T, nz = wino_geom(h, w, d, m), (m + 2) ** 2
            k = "wino_" if m == 2 else "wino4_"
            add(k + "input_kernel", B * (h * w * cin + nz * T * cin) * 4)
            add(k + "output_kernel", B * (nz * T * cout + h * w * cout * (2 if res else 1)) * 4)
        add("psp_pool_kernel", B * (h * w * 512 + 50 * 512) * 4)
        add("psp_prior_sum_kernel", B * (50 * 1024 + h * w * 1024) * 4)
        add("upconv_gather_tiled_kernel", B * (h * w * 9 * 256 + 4 * h * w * 256) * 4)          # up_1
        add("upconv_gather_tiled_kernel", B * (4 * h * w * 9 * 64 + 16 * h * w * 64) * 4)        # up_2
        add("up3_patch_kernel", B * (min(16 * h * w * 64, N * 36 * 64) + NPAD * 576) * 4)
    Bt = PER * len(bench.CROPS)
    add("final_lsm_kernel", Bt * (NPAD * 64 + 2 * N * 32) * 4)
    add("cloud_conv1_kernel", (1 + ITERS) * Bt * (N * 3 + NPAD * 64) * 4)
    add("fc_rows_kernel", (1920 * 1024 + Bt * (1024 + 1920)) * 4)                                      # global-feature half of head layer 1
    add("fc_rows_kernel", ITERS * ((1024 * 1024 + Bt * 2048) + (2 * 128 * 512 + Bt * (1024 + 256))) * 4)   # refiner FC towers
    add("head_conf_kernel", Bt * (NPAD * 128 + N) * 4)
    add("head_select_kernel", (Bt * (N + 384 + 1920) + 1280 * 384 + 2 * 256 * 640 + 2 * 128 * 256) * 4)
    add("colsum_finish_kernel", (1 + ITERS) * Bt * (16 * 1024 + 1024) * 4)
    return b


def main():
    path, out = sys.argv[1], sys.argv[2]
    alg = step_bytes()
    alg["wino4_input_multi_kernel"] = alg.pop("wino4_input_kernel", 0.0)          # the F(4x4) transforms run all buckets in one launch
    alg["wino4_output_multi_kernel"] = alg.pop("wino4_output_kernel", 0.0)
    last = json.load(open(path))
    rows = []
    for r in last["kernels"]:
        k = r["kernel"]
        if k in alg and alg[k] > 0:
            us = r["us_per_step"]
            gbs = alg[k] / us / 1e3
            rows.append({"kernel": k, "calls_per_step": r["calls_per_step"], "us_per_step": us, "algorithmic_mb_per_step": round(alg[k] / 1e6, 1),
                         "achieved_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM, 3)})
    rows.sort(key=lambda r: -r["us_per_step"])
    res = {"workload": "bench.py step: 280 objects (40 of each of 7 crop sizes), N=1000, 2 refine iterations; serial run (one stream, no graph, --groups 1); "
                       "steady state = the LAST pass of the trace (tools/trace_summary.py)",
           "hbm_peak_GBps": HBM, "source": path, "kernels": rows, "glue_us_per_step": round(sum(r["us_per_step"] for r in rows), 1)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for r in rows:
        print(f"{r['kernel']:30s} {r['us_per_step']:8.1f} us  {r['algorithmic_mb_per_step']:9.1f} MB  {r['achieved_GBps']:7.1f} GB/s  {r['hbm_frac']:.3f}")
    print("glue total", res["glue_us_per_step"], "us / step")


if __name__ == "__main__":
    main()
