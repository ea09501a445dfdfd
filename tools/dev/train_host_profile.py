"""Dev tool: cProfile of tools/train.py's host loop on the synthetic dataset (one phase), top entries by cumulative time."""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import train as train_tool

if __name__ == "__main__":
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        argv = ["--dataset", "synthetic", "--synthetic_train_frames", "1024", "--synthetic_test_frames", "8", "--outf", d + "/m", "--log_dir", d + "/l",
                "--decay_margin", "0", "--refine_margin", "0"] + extra
        train_tool.main(argv + ["--nepoch", "2"])
        pr = cProfile.Profile()
        pr.enable()
        train_tool.main(argv + ["--nepoch", "4"])
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(45)
