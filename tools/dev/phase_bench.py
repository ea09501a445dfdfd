"""Dev tool: frames/s of tools/train.py's two phases on the synthetic dataset (YCB shape: K=21, N=1000), one epoch each after a
warm-up epoch: PoseNet phase and refiner phase (--refine_start: frozen estimator through the inference engine + `iteration` native
refiner steps per frame).  usage: phase_bench.py [extra train.py flags]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import train as train_tool

if __name__ == "__main__":
    import tempfile
    extra = sys.argv[1:]
    NF = 1024                     # training frames per epoch (this fork's default --batch_size 32: 32 windows; 64 refiner windows of 16)
    for phase in ("posenet", "refiner"):
        with tempfile.TemporaryDirectory() as d:
            argv = ["--dataset", "synthetic", "--synthetic_train_frames", str(NF), "--synthetic_test_frames", "8", "--outf", d + "/m", "--log_dir", d + "/l",
                    "--decay_margin", "0", "--refine_margin", "0"] + (["--refine_start"] if phase == "refiner" else []) + extra
            train_tool.main(argv + ["--nepoch", "2"])                                       # warm-up: library, workspaces, worker start
            t0 = time.perf_counter(); train_tool.main(argv + ["--nepoch", "3"]); t1 = time.perf_counter()     # 2 epochs (start_epoch = 1 .. nepoch - 1)
            train_tool.main(argv + ["--nepoch", "9"]); t2 = time.perf_counter()                                # 8 epochs
            one = ((t2 - t1) - (t1 - t0)) / 6
            print(f"{phase}: {NF / one:.1f} frames/s ({NF} training frames + 8 test frames in {one:.2f} s)", flush=True)
