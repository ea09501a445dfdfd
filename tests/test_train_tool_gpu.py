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
    """--dataset linemod: the built-in loader (device-side preparation, the reference's training augmentation) feeds the
    trainer; one epoch over a fabricated tree runs and reports a finite test distance."""
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


def test_resumed_refiner_run_is_the_refiner_phase_from_the_start(tmp_path, monkeypatch, caplog):
    """--resume_refinenet (tools/train.py:86-100 of the reference): refine_start is set BEFORE the datasets are built (YCB
    then samples 2600 mesh points), and the lr / w decay is not applied a second time."""
    import logging
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    from densefusion_amd import synth
    common = ["--dataset", "synthetic", "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "8",
              "--synthetic_test_frames", "2", "--batch_size", "4", "--outf", str(tmp_path / "models"), "--log_dir", str(tmp_path / "logs")]
    train.SyntheticPoseDataset.CROPS = [(40, 40)]
    os.makedirs(tmp_path / "models", exist_ok=True)
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(2), 5).items()}, tmp_path / "models" / "p.pth")
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(2), 6).items()}, tmp_path / "models" / "r.pth")
    seen = []
    orig = train.make_datasets
    monkeypatch.setattr(train, "make_datasets", lambda opt: (seen.append((opt.refine_start, opt.batch_size)), orig(opt))[1])
    with caplog.at_level(logging.INFO, logger="train"):
        train.main(common + ["--nepoch", "3", "--resume_posenet", "p.pth", "--resume_refinenet", "r.pth", "--decay_margin", "1e9",
                             "--refine_margin", "1e9"])
    assert seen and seen[0] == (True, 2)                      # refiner phase (and batch_size / iteration) known when the datasets are built
    assert not [r for r in caplog.records if r.getMessage().startswith("decay:")]       # already decayed: never again


def test_two_rank_trainer_with_a_dataset_that_does_not_divide(tmp_path):
    """Two data-parallel ranks (gloo, both on the one card) over 15 frames with 4 frames per optimizer step: the shards come
    from one shared permutation and every rank takes the same number of steps, so no rank is left waiting in the gradient
    all-reduce (the run finishes) and both log the same number of optimizer steps."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DF_TRAIN_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "train.py"), "--dist_backend", "gloo", "--dataset", "synthetic",
           "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "15", "--synthetic_test_frames", "3", "--batch_size", "4",
           "--nepoch", "3", "--refine_margin", "-1", "--decay_margin", "-1", "--outf", str(tmp_path / "models"), "--log_dir", str(tmp_path / "logs")]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-3000:]
    # rank 0 logs at INFO: 15 // (2 * 4) = 1 optimizer step per epoch, two epochs
    assert out.stdout.count("Train time") == 2 and out.stdout.count("TEST FINISH") == 2, out.stdout[-3000:]


def test_two_rank_resume_of_the_estimator_alone_keeps_the_refiners_in_sync(tmp_path):
    """--resume_posenet without --resume_refinenet on two ranks (the usual way into the refiner phase): every rank loads the same
    PoseNet but builds its own refiner under a per-rank seed, so the fresh network must still be broadcast -- the trainer checks the
    replicas (train_utils.replicas_in_sync) and refuses to start otherwise."""
    import socket
    import subprocess
    from densefusion_amd import synth
    os.makedirs(tmp_path / "models", exist_ok=True)
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(2), 5).items()}, tmp_path / "models" / "p.pth")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DF_TRAIN_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "train.py"), "--dist_backend", "gloo", "--dataset", "synthetic",
           "--num_objects", "2", "--num_points", "64", "--synthetic_train_frames", "8", "--synthetic_test_frames", "2", "--batch_size", "2",
           "--nepoch", "2", "--resume_posenet", "p.pth", "--refine_margin", "1e9", "--decay_margin", "-1", "--outf", str(tmp_path / "models"),
           "--log_dir", str(tmp_path / "logs")]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "different weights" not in out.stdout and out.stdout.count("TEST FINISH") >= 1, out.stdout[-3000:]


def test_refiner_phase_as_one_window_and_on_lanes_matches_one_frame_at_a_time(tmp_path):
    """The refiner phase three ways: one frame at a time (--passes lanes --lanes 1, the reference's bs = 1 loop), the frames of a window on 3
    lanes (own refiner step AND own copy of the frozen estimator per lane), and the default --passes window (the frozen estimator over the
    window's mixed crop sizes in ONE multi-bucket forward, then the refiner steps over all frames at once).  Same data order, the same
    refiner after an epoch: a different summation order of the window's gradient only (tolerance, not bits)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    from densefusion_amd import synth
    train.SyntheticPoseDataset.CROPS = [(40, 40), (80, 40)]
    os.makedirs(tmp_path / "ck", exist_ok=True)
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(3), 5).items()}, tmp_path / "ck" / "p.pth")
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(3), 6).items()}, tmp_path / "ck" / "r.pth")
    got = {}
    for name, flags in (("one", ["--passes", "lanes", "--lanes", "1"]), ("lanes3", ["--passes", "lanes", "--lanes", "3"]), ("window", ["--passes", "window"])):
        out = tmp_path / f"m_{name}"
        os.makedirs(out)
        for f in ("p.pth", "r.pth"):
            os.link(tmp_path / "ck" / f, out / f)
        train.main(["--dataset", "synthetic", "--num_objects", "3", "--num_points", "64", "--synthetic_train_frames", "24", "--synthetic_test_frames", "2",
                    "--batch_size", "12", "--nepoch", "2", "--resume_posenet", "p.pth", "--resume_refinenet", "r.pth", "--decay_margin", "1e9",
                    "--refine_margin", "1e9", "--outf", str(out), "--log_dir", str(tmp_path / f"l_{name}")] + flags)
        ck = glob.glob(str(out / "pose_refine_model_1_*.pth"))
        assert ck, os.listdir(out)
        got[name] = torch.load(ck[0], map_location="cpu", weights_only=True)
    moved = 0.0
    start = torch.load(tmp_path / "ck" / "r.pth", weights_only=True)
    for k in got["one"]:
        a = got["one"][k].double()
        moved = max(moved, float((a - start[k].double()).abs().max()))
        for other in ("lanes3", "window"):
            assert float((a - got[other][k].double()).abs().max()) <= 2e-5 + 2e-3 * float((a - start[k].double()).abs().max()), (other, k)
    assert moved > 1e-5                                                                      # it trained


def test_a_window_larger_than_window_pixels_is_cut_into_passes(tmp_path, caplog):
    """--window_pixels bounds the workspace: a window whose crops add up to more pixels runs as several multi-bucket passes (PoseNet phase) /
    several estimator forwards (refiner phase); the optimizer still steps once per window and both phases produce sane distances."""
    import logging
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train
    from densefusion_amd import synth
    train.SyntheticPoseDataset.CROPS = [(40, 40), (80, 40), (40, 80)]
    os.makedirs(tmp_path / "m", exist_ok=True)
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(3), 5).items()}, tmp_path / "m" / "p.pth")
    common = ["--dataset", "synthetic", "--num_objects", "3", "--num_points", "64", "--synthetic_train_frames", "24", "--synthetic_test_frames", "2",
              "--batch_size", "12", "--nepoch", "3", "--resume_posenet", "p.pth", "--decay_margin", "-1", "--outf", str(tmp_path / "m"),
              "--log_dir", str(tmp_path / "l"), "--window_pixels", "4000"]                     # 1-2 frames per pass
    for extra in (["--refine_margin", "-1"], ["--refine_margin", "1e9"]):                       # PoseNet phase only / refiner phase from epoch 2
        caplog.clear()
        with caplog.at_level(logging.INFO, logger="train"):
            best = train.main(common + extra)
        lines = [r.getMessage() for r in caplog.records]
        batches = [float(ln.split("Avg_dis:")[1]) for ln in lines if ln.startswith("Train time") and "Batch" in ln]
        assert best == best and 1e-5 < best < 10 and len(batches) >= 4 and all(0 < b < 10 for b in batches), (best, batches)
