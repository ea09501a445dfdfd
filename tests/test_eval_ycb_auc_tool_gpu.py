"""GPU: tools/eval_ycb_auc.py (the numeric job of evaluate_poses_keyframe.m + plot_accuracy_keyframe.m) on a fabricated
toolbox / dataset tree, against the numpy restatement of those scripts (oracle/ycb_metric.py)."""
import json
import os
import sys

import numpy as np
import pytest
import scipy.io as scio

from densefusion_amd import synth
from densefusion_amd.lib.transformations import quaternion_from_matrix
from oracle import ycb_metric

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rt(rng, scale=1.0):
    R = synth.quat_to_rot(synth.random_unit_quaternion(rng))
    return np.concatenate([R, (rng.normal(size=(3, 1)) * 0.3 + [[0], [0], [0.9]]) * scale], axis=1)


def test_eval_ycb_auc_entry_point(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_ycb_auc
    rng = np.random.default_rng(11)
    classes = ["002_master_chef_can", "003_cracker_box", "004_sugar_box"]
    root, tool = tmp_path / "YCB", tmp_path / "toolbox"
    os.makedirs(tool / "results_PoseCNN_RSS2018"); os.makedirs(tmp_path / "ref"); os.makedirs(tmp_path / "wo")
    (tool / "classes.txt").write_text("\n".join(classes) + "\n")
    models = []
    for c in classes:
        os.makedirs(root / "models" / c)
        pts = (rng.random((300, 3)) - 0.5) * 0.15
        np.savetxt(root / "models" / c / "points.xyz", pts, fmt="%.6f")
        models.append(np.loadtxt(root / "models" / c / "points.xyz"))
    keyframes = ["0048/000001", "0048/000036", "0050/000007", "0050/000120", "0051/000002"]
    (tool / "keyframe.txt").write_text("\n".join(keyframes) + "\n")
    expect = []                                               # (cls, add_ref, adi_ref, add_wo, adi_wo) per gt instance
    for i, name in enumerate(keyframes):
        seq, frame = name.split("/")
        os.makedirs(root / "data" / seq, exist_ok=True)
        present = [1, 2, 3] if i % 2 == 0 else [3, 1]
        gt = np.stack([_rt(rng) for _ in present], axis=2)
        scio.savemat(root / "data" / seq / f"{frame}-meta.mat", {"cls_indexes": np.array(present, dtype=np.float64)[:, None], "poses": gt})
        det_cls = [c for c in present if not (i == 1 and c == 3)]          # one missed detection
        if i == 3:
            det_cls = det_cls[::-1]                                         # detection order differs from the gt order
        rois = np.array([[0, c, 10, 10, 100, 100, 0.9] for c in det_cls], dtype=np.float64)
        scio.savemat(tool / "results_PoseCNN_RSS2018" / f"{i:06d}.mat", {"rois": rois})
        ref_rows, wo_rows = [], []
        for c in det_cls:
            g = gt[:, :, present.index(c)]
            rows = []
            for noise in (0.004, 0.03):                                     # refined: close; without refinement: coarser
                dR = synth.quat_to_rot(synth.random_unit_quaternion(rng))
                R = g[:, :3] @ (np.eye(3) * (1 - noise * 3) + dR * noise * 3)
                u, _, vt = np.linalg.svd(R)
                R = u @ vt
                t = g[:, 3] + rng.normal(size=3) * noise
                M4 = np.eye(4); M4[:3, :3] = R
                rows.append(np.concatenate([quaternion_from_matrix(M4, True), t]))
            if i == 4 and c == 1:
                rows[0] = np.zeros(7)                                       # tools/eval_ycb.py writes zeros for a lost object
            ref_rows.append(rows[0]); wo_rows.append(rows[1])
        scio.savemat(tmp_path / "ref" / f"{i:04d}.mat", {"poses": np.array(ref_rows)})
        scio.savemat(tmp_path / "wo" / f"{i:04d}.mat", {"poses": np.array(wo_rows)})
        for j, c in enumerate(present):
            vals = [c]
            for rows_all in (ref_rows, wo_rows):
                if c in det_cls and np.any(rows_all[det_cls.index(c)]):
                    from densefusion_amd.lib.ycb_eval import pose_to_rt
                    rt = pose_to_rt(rows_all[det_cls.index(c)])
                    vals += [ycb_metric.add(rt, gt[:, :, j], models[c - 1].T), ycb_metric.adi(rt, gt[:, :, j], models[c - 1].T)]
                else:
                    vals += [np.inf, np.inf]
            expect.append(vals)
    table = eval_ycb_auc.main(["--dataset_root", str(root), "--ycb_toolbox_dir", str(tool), "--result_refine_dir", str(tmp_path / "ref"),
                               "--result_wo_refine_dir", str(tmp_path / "wo"), "--output_dir", str(tmp_path / "out")])
    res = scio.loadmat(tmp_path / "out" / "results_keyframe.mat")
    E = np.array(expect)
    assert res["distances_sys"].shape == (len(expect), 5) and np.array_equal(res["results_cls_id"][:, 0], E[:, 0])
    for col, (ca, cs) in ((0, (1, 2)), (2, (3, 4))):
        for got, want in ((res["distances_non"][:, col], E[:, ca]), (res["distances_sys"][:, col], E[:, cs])):
            assert np.array_equal(np.isinf(got), np.isinf(want))
            np.testing.assert_allclose(got[np.isfinite(got)], want[np.isfinite(want)], rtol=1e-10, atol=1e-12)
    assert np.isinf(res["distances_sys"][:, [1, 3, 4]]).all()
    for k, name in enumerate(classes + ["All 3 objects"]):
        sel = E[:, 0] == k + 1 if k < 3 else np.ones(len(E), dtype=bool)
        for tag, (ca, cs) in (("iterative", (1, 2)), ("per-pixel", (3, 4))):
            auc_s, lt_s = ycb_metric.auc_and_lt2cm(E[sel, cs])
            auc_n, lt_n = ycb_metric.auc_and_lt2cm(E[sel, ca])
            row = table[name][tag]
            assert abs(row["ADD-S_AUC"] - auc_s * 100) < 1e-9 and abs(row["ADD_AUC"] - auc_n * 100) < 1e-9
            assert abs(row["ADD-S_lt2cm"] - lt_s * 100) < 1e-9 and abs(row["ADD_lt2cm"] - lt_n * 100) < 1e-9
    assert table["All 3 objects"]["iterative"]["ADD-S_AUC"] > table["All 3 objects"]["per-pixel"]["ADD-S_AUC"] > 0
    assert json.load(open(tmp_path / "out" / "accuracy.json")).keys() == table.keys()
