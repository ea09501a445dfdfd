"""GPU: tools/eval_linemod.py end to end with an injected synthetic dataset object (the LineMOD loader is
dataset tooling outside this build): distances and pass/fail decisions must match the oracle pipeline."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

from densefusion_amd import synth
from oracle import dfnet, pose_math

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeLinemod:
    def __init__(self, n, num_pt):
        self.items = [synth.make_object(4000 + i, 80, 80, num_pt, 13, 500, cam=synth.LINEMOD_CAM) for i in range(n)]
        for i, o in enumerate(self.items):
            o["obj"][0] = [0, 7, 8, 3][i % 4]          # objects 7, 8 (eggbox, glue) are the symmetric ones

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        o = self.items[i]
        return (torch.from_numpy(o["cloud"]), torch.from_numpy(o["choose"]), torch.from_numpy(o["img"]),
                torch.from_numpy(o["target"]), torch.from_numpy(o["model_points"]), torch.from_numpy(o["obj"]))

    def get_sym_list(self):
        return [7, 8]

    def get_num_points_mesh(self):
        return 500


def test_eval_linemod_entry_point(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_linemod
    K, N = 13, 500
    sdp, sdr = synth.make_state_dict(synth.posenet_spec(K), 31), synth.make_state_dict(synth.refiner_spec(K), 1031)
    torch.save({k: torch.from_numpy(v) for k, v in sdp.items()}, tmp_path / "p.pth")
    torch.save({k: torch.from_numpy(v) for k, v in sdr.items()}, tmp_path / "r.pth")
    os.makedirs(tmp_path / "cfg")
    yaml.safe_dump({o: {"diameter": 400.0 + 10 * o} for o in eval_linemod.OBJLIST}, open(tmp_path / "cfg" / "models_info.yml", "w"))
    ds = FakeLinemod(6, N)
    succ, cnt = eval_linemod.main(["--model", str(tmp_path / "p.pth"), "--refine_model", str(tmp_path / "r.pth"), "--dataset_config_dir",
                                   str(tmp_path / "cfg"), "--output_result_dir", str(tmp_path / "out")], testdataset=ds)
    log = open(tmp_path / "out" / "eval_result_logs.txt").read().splitlines()
    assert sum(cnt) == 6 and log[-1].startswith("ALL success rate")
    tp, tr = dfnet._to_torch_sd(sdp), dfnet._to_torch_sd(sdr)
    for i in range(6):
        o = ds.items[i]
        with torch.no_grad():
            _, pose = pose_math.estimate_pose(tp, tr, torch.from_numpy(o["img"])[None], torch.from_numpy(o["cloud"])[None],
                                              torch.from_numpy(o["choose"]), torch.from_numpy(o["obj"]), 4)
        pred = pose_math.transform_model(pose, o["model_points"])
        want = pose_math.adds_metric(pred, o["target"]) if int(o["obj"][0]) in (7, 8) else pose_math.add_metric(pred, o["target"])
        got = float(log[i].split("Distance: ")[1])
        assert abs(got - want) < 1e-4, (i, got, want)


def test_eval_linemod_log_does_not_depend_on_the_window(tmp_path):
    """--window W (crops of W frames bucketed by size, one device call) writes the log of the frame-by-frame run, line for line;
    mixed crop sizes and a lost-detection sentinel included."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_linemod
    K, N = 13, 500
    sdp, sdr = synth.make_state_dict(synth.posenet_spec(K), 31), synth.make_state_dict(synth.refiner_spec(K), 1031)
    torch.save({k: torch.from_numpy(v) for k, v in sdp.items()}, tmp_path / "p.pth")
    torch.save({k: torch.from_numpy(v) for k, v in sdr.items()}, tmp_path / "r.pth")
    os.makedirs(tmp_path / "cfg")
    yaml.safe_dump({o: {"diameter": 400.0 + 10 * o} for o in eval_linemod.OBJLIST}, open(tmp_path / "cfg" / "models_info.yml", "w"))

    class Mixed(FakeLinemod):
        def __init__(self):
            sizes = [(80, 80), (120, 160), (80, 80), (120, 120), (120, 160), (80, 80), (160, 160)]
            self.items = [synth.make_object(4100 + i, H, W, N, 13, 500, cam=synth.LINEMOD_CAM) for i, (H, W) in enumerate(sizes)]
            for i, o in enumerate(self.items):
                o["obj"][0] = [0, 7, 8, 3][i % 4]

        def __getitem__(self, i):
            if i == 3:                                  # datasets/linemod/dataset.py:135-137: six LongTensor([0])
                return tuple(torch.LongTensor([0]) for _ in range(6))
            return super().__getitem__(i)

    logs = {}
    for window in (1, 5):
        out = tmp_path / f"out{window}"
        eval_linemod.main(["--model", str(tmp_path / "p.pth"), "--refine_model", str(tmp_path / "r.pth"), "--dataset_config_dir",
                           str(tmp_path / "cfg"), "--output_result_dir", str(out), "--window", str(window)], testdataset=Mixed())
        logs[window] = open(out / "eval_result_logs.txt").read()
    assert logs[1] == logs[5] and "No.3 NOT Pass! Lost detection!" in logs[1] and logs[1].count("Distance:") == 6
