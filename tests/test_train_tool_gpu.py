"""GPU: tools/train.py end to end on seeded synthetic frames (tiny config): both phases run, the loss goes
down, checkpoints are written with the reference's names / key layout and load back into the eval path."""
import glob
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_entry_point_both_phases(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    from densefusion_amd import synth
    common = ["--dataset", "synthetic", "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "16",
              "--synthetic_test_frames", "4", "--batch_size", "4", "--lr", "0.0005", "--outf", str(tmp_path / "models"),
              "--log_dir", str(tmp_path / "logs")]
    train.SyntheticPoseDataset.CROPS = [(40, 40), (40, 80)]
    first = train.main(common + ["--nepoch", "2", "--refine_margin", "-1", "--decay_margin", "-1"])
    best = train.main(common + ["--nepoch", "5", "--refine_margin", "-1", "--decay_margin", "-1"])
    assert best == best and best < first + 1e-6                      # finite and not worse after more epochs
    ckpts = sorted(glob.glob(str(tmp_path / "models" / "pose_model_*.pth")))
    assert ckpts, "no PoseNet checkpoint written"
    sd = torch.load(ckpts[-1], weights_only=True)
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == synth.posenet_spec(2)
    # phase B: refiner training starting from that checkpoint (refine_margin = +inf switches immediately)
    name = os.path.basename(ckpts[-1])
    train.main(common + ["--nepoch", "3", "--refine_margin", "1e9", "--decay_margin", "-1", "--resume_posenet", name])
    assert glob.glob(str(tmp_path / "models" / "pose_refine_model_*.pth")), "no refiner checkpoint written"


def test_train_entry_point_shared_passes(tmp_path):
    """--frames_per_pass: same-size frames of an accumulation window share a pass; both phases still run and learn."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    common = ["--dataset", "synthetic", "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "16",
              "--synthetic_test_frames", "4", "--batch_size", "4", "--frames_per_pass", "4", "--lr", "0.0005",
              "--outf", str(tmp_path / "models"), "--log_dir", str(tmp_path / "logs")]
    train.SyntheticPoseDataset.CROPS = [(40, 40), (40, 80)]
    first = train.main(common + ["--nepoch", "2", "--refine_margin", "-1", "--decay_margin", "-1"])
    best = train.main(common + ["--nepoch", "5", "--refine_margin", "-1", "--decay_margin", "-1"])
    assert best == best and best < first + 1e-6
    ckpts = sorted(glob.glob(str(tmp_path / "models" / "pose_model_*.pth")))
    train.main(common + ["--nepoch", "3", "--refine_margin", "1e9", "--decay_margin", "-1", "--resume_posenet", os.path.basename(ckpts[-1])])
    assert glob.glob(str(tmp_path / "models" / "pose_refine_model_*.pth"))
