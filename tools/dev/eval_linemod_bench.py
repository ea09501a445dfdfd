"""Dev tool: frames/s of tools/eval_linemod.py on a fabricated LineMOD tree (PNG decoding, segmentation-label boxes, device
preparation, pose estimation + metric per window) with the frames fetched in the loop, by worker threads and by worker processes.
usage: eval_linemod_bench.py [frames_per_object]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import contextlib, io
import torch, yaml

if __name__ == "__main__":
    import eval_linemod
    from densefusion_amd import synth
    from test_linemod_dataset_gpu import make_tree
    per_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    with tempfile.TemporaryDirectory() as d:
        tree = make_tree(d + "/lm", frames_per_obj=per_obj)
        torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(13), 5).items()}, d + "/p.pth")
        torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(13), 6).items()}, d + "/r.pth")
        os.makedirs(d + "/cfg")
        yaml.safe_dump({o: {"diameter": 400.0} for o in eval_linemod.OBJLIST}, open(d + "/cfg/models_info.yml", "w"))
        for name, extra in (("in the loop", ["--workers", "0"]), ("8 threads", ["--workers", "8", "--feed", "threads"]), ("8 processes", ["--workers", "8"]),
                            ("16 processes", ["--workers", "16"])):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                succ, cnt = eval_linemod.main(["--model", d + "/p.pth", "--refine_model", d + "/r.pth", "--dataset_root", tree, "--dataset_config_dir", d + "/cfg",
                                               "--output_result_dir", d + "/out"] + extra)
            dt = time.perf_counter() - t0
            print(f"{name}: {13 * per_obj / dt:.1f} frames/s ({13 * per_obj} frames in {dt:.2f} s incl. start-up)", flush=True)
