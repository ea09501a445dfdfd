"""GPU: tools/train.py on a fabricated LineMOD tree, fed by worker processes (--feed processes, the default for the disk datasets) and
by threads: both runs go through the same epochs (optimizer steps, per-epoch test pass, checkpoint) and the worker processes of the
first epoch serve the second."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("feed", ["processes", "threads"])
def test_train_tool_on_a_linemod_tree(tmp_path, feed):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_linemod_dataset_gpu import make_tree
    tree = make_tree(str(tmp_path / "tree"), frames_per_obj=12)
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train.py"), "--dataset", "linemod", "--dataset_root", tree, "--nepoch", "3", "--repeat_epoch", "1",
           "--batch_size", "4", "--workers", "3", "--feed", feed, "--lanes", "2", "--outf", str(out / "models"), "--log_dir", str(out / "logs"),
           "--decay_margin", "0", "--refine_margin", "0"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    log = r.stdout + r.stderr
    assert log.count("train finish") == 2, log[-3000:]                     # epochs 1 and 2
    assert any(f.startswith("pose_model_") for f in os.listdir(out / "models")), os.listdir(out / "models")


def test_train_tool_on_a_ycb_tree_with_synthetic_frames(tmp_path):
    """tools/train.py --dataset ycb on a fabricated YCB-Video tree: real and data_syn frames, the training augmentation (occluders,
    backgrounds, jitter, noise) in worker processes, two epochs, a checkpoint."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_ycb_dataset_gpu import make_tree
    root, cfg = str(tmp_path / "YCB"), str(tmp_path / "cfg")
    make_tree(root, cfg, np.random.default_rng(9))
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train.py"), "--dataset", "ycb", "--dataset_root", root, "--dataset_config_dir", cfg, "--nepoch", "3",
           "--batch_size", "3", "--workers", "2", "--lanes", "2", "--outf", str(out / "models"), "--log_dir", str(out / "logs"), "--decay_margin", "0",
           "--refine_margin", "0"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    log = r.stdout + r.stderr
    assert log.count("train finish") == 2 and "training augmentation on" in log, log[-3000:]
    assert any(f.startswith("pose_model_") for f in os.listdir(out / "models")), os.listdir(out / "models")
    logs = sorted(os.listdir(out / "logs"))                                  # the reference's per-epoch log files (lib/utils.py)
    assert logs == ["epoch_1_log.txt", "epoch_1_test_log.txt", "epoch_2_log.txt", "epoch_2_test_log.txt"], logs
    assert "Avg_dis" in open(out / "logs" / "epoch_1_log.txt").read() and "TEST FINISH" in open(out / "logs" / "epoch_2_test_log.txt").read()
