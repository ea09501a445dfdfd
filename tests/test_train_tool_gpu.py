"""GPU: tools/train.py end to end on seeded synthetic frames (tiny config): both phases run, the loss goes
down, checkpoints are written with the reference's names / key layout and load back into the eval path."""
import glob
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eval_on_train_frames(train, monkeypatch):
    """The synthetic images carry no information about the pose, so a held-out set cannot show learning; the per-epoch
    test pass of this unit test therefore runs on the training frames (memorisation is what 16 frames can show)."""
    orig = train.make_datasets
    monkeypatch.setattr(train, "make_datasets", lambda opt: (orig(opt)[0],) * 2)


def _phase_b(train, common, posenet_ckpt, caplog):
    """Refiner phase from a PoseNet checkpoint: refine_margin = +inf switches after the first test pass; the following
    epochs train the refiner (its distances are logged) -- a refiner checkpoint is written only when the refined distance
    beats the best so far, as in the reference (tools/train.py:205-213), so it is not required here."""
    import logging
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="train"):
        got = train.main(common + ["--nepoch", "5", "--refine_margin", "1e9", "--decay_margin", "-1", "--resume_posenet", posenet_ckpt])
    lines = [r.getMessage() for r in caplog.records]
    batches = [ln for ln in lines if ln.startswith("Train time") and "Batch" in ln]
    tests = [float(ln.split("Avg dis: ")[1]) for ln in lines if "TEST FINISH" in ln]
    assert got == got and len(tests) == 4 and all(1e-5 < t < 10 for t in tests)
    assert len(batches) >= 4 + 3 * 8            # epoch 1: 4 PoseNet windows of 4; epochs 2-4: refiner windows of 4 / iteration = 2


def test_train_entry_point_both_phases(tmp_path, monkeypatch, caplog):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    from densefusion_amd import synth
    common = ["--dataset", "synthetic", "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "16",
              "--synthetic_test_frames", "4", "--batch_size", "4", "--outf", str(tmp_path / "models"),
              "--log_dir", str(tmp_path / "logs"), "--lr", "0.0002"]
    train.SyntheticPoseDataset.CROPS = [(40, 40), (40, 80)]
    _eval_on_train_frames(train, monkeypatch)
    untrained = train.main(common + ["--nepoch", "2", "--refine_margin", "-1", "--decay_margin", "-1", "--lr", "0"])
    best = train.main(common + ["--nepoch", "5", "--refine_margin", "-1", "--decay_margin", "-1"])
    assert 1e-4 < untrained < 10 and best == best and 1e-4 < best < 0.9 * untrained       # a real distance, and it learns
    ckpts = sorted(glob.glob(str(tmp_path / "models" / "pose_model_*.pth")))
    assert ckpts, "no PoseNet checkpoint written"
    sd = torch.load(ckpts[-1], weights_only=True)
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == synth.posenet_spec(2)
    # phase B: refiner training starting from that checkpoint (refine_margin = +inf switches immediately)
    name = os.path.basename(ckpts[-1])
    _phase_b(train, common, name, caplog)


def test_train_entry_point_shared_passes(tmp_path, monkeypatch, caplog):
    """--frames_per_pass: same-size frames of an accumulation window share a pass; both phases still run and learn."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    common = ["--dataset", "synthetic", "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "16",
              "--synthetic_test_frames", "4", "--batch_size", "4", "--frames_per_pass", "4",
              "--outf", str(tmp_path / "models"), "--log_dir", str(tmp_path / "logs"), "--lr", "0.0002"]
    train.SyntheticPoseDataset.CROPS = [(40, 40), (40, 80)]
    _eval_on_train_frames(train, monkeypatch)
    untrained = train.main(common + ["--nepoch", "2", "--refine_margin", "-1", "--decay_margin", "-1", "--lr", "0"])
    best = train.main(common + ["--nepoch", "5", "--refine_margin", "-1", "--decay_margin", "-1"])
    assert 1e-4 < untrained < 10 and best == best and 1e-4 < best < 0.9 * untrained
    ckpts = sorted(glob.glob(str(tmp_path / "models" / "pose_model_*.pth")))
    _phase_b(train, common, os.path.basename(ckpts[-1]), caplog)


def test_train_on_fabricated_linemod_tree_with_builtin_loader(tmp_path, caplog):
    """--dataset linemod without the reference's loader on the PYTHONPATH: the built-in loader (device-side preparation, no
    augmentation) feeds the trainer; one epoch over a fabricated tree runs and reports a finite test distance."""
    import logging
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    from test_linemod_dataset_gpu import make_tree
    tree = make_tree(str(tmp_path / "lm"), frames_per_obj=10)          # 'test' keeps line 10 of every list: 13 test frames
    with caplog.at_level(logging.INFO, logger="train"):
        best = train.main(["--dataset", "linemod", "--dataset_root", tree, "--nepoch", "2", "--batch_size", "8", "--frames_per_pass", "4",
                           "--refine_margin", "-1", "--decay_margin", "-1", "--outf", str(tmp_path / "models"), "--log_dir", str(tmp_path / "logs")])
    lines = [r.getMessage() for r in caplog.records]
    assert any("built-in loader" in ln for ln in lines)
    assert sum(ln.startswith("Train time") for ln in lines) >= 10 and 1e-4 < best < 10
    assert glob.glob(str(tmp_path / "models" / "pose_model_1_*.pth"))
