"""Dev tool: keyframes/s and poses/s of tools/eval_ycb.py on a fabricated YCB-Video tree (noise PNGs, PoseCNN-style .mat
detections, 2 objects per frame), reader threads 0 / 8 / 16.  usage: eval_ycb_bench.py [frames]"""
import contextlib, io, os, pathlib, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if __name__ == "__main__":
    import eval_ycb
    from test_eval_ycb_tool_gpu import _fabricate
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    with tempfile.TemporaryDirectory() as d:
        tmp = pathlib.Path(d)
        root, toolbox, cfg, frames, _, _ = _fabricate(tmp, np.random.default_rng(5), n, 21, 1000)
        for workers in (1, 8, 16):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                eval_ycb.main(["--dataset_root", str(root), "--model", str(tmp / "pose_model.pth"), "--refine_model", str(tmp / "pose_refine_model.pth"),
                               "--dataset_config_dir", str(cfg), "--ycb_toolbox_dir", str(toolbox), "--result_wo_refine_dir", str(tmp / "wo"),
                               "--result_refine_dir", str(tmp / "ref"), "--workers", str(workers)])
            dt = time.perf_counter() - t0
            print(f"reader threads {workers}: {n / dt:.1f} keyframes/s, {2 * n / dt:.1f} poses/s ({dt:.2f} s incl. start-up)", flush=True)
