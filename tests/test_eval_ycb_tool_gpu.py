"""GPU: the tools/eval_ycb.py entry point end to end on a fabricated miniature YCB-Video tree (no dataset
exists offline): PNG frames, PoseCNN-style .mat detections, .pth checkpoints in the reference's key layout ->
result .mat files, compared with the oracle pipeline (numpy input preparation + CPU network + host loop)."""
import os
import sys

import numpy as np
import pytest
import scipy.io as scio
import torch
from PIL import Image

from densefusion_amd import synth
from oracle import dfnet, pose_math, preprocess_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fabricate(tmp, rng, n_frames, K, N):
    root, tool, cfg = tmp / "YCB", tmp / "toolbox" / "results_PoseCNN_RSS2018", tmp / "cfg"
    for d in (root / "data" / "0001", tool, cfg):
        os.makedirs(d, exist_ok=True)
    names = [f"data/0001/{i:06d}" for i in range(n_frames)]
    (cfg / "test_data_list.txt").write_text("\n".join(names) + "\n")
    frames = []
    for fi, nm in enumerate(names):
        rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        depth = rng.integers(5000, 12000, (480, 640)).astype(np.uint16)
        depth[rng.random((480, 640)) < 0.05] = 0
        label = np.zeros((480, 640), dtype=np.uint8)
        rois = []
        for it, (r0, c0, h, w) in zip((2, 7), ((40, 60, 75, 110), (250, 300, 150, 150))):
            label[r0:r0 + h, c0:c0 + w][rng.random((h, w)) < 0.8] = it
            rois.append([0, it, c0, r0, c0 + w, r0 + h, 0.9])
        Image.fromarray(rgb).save(root / f"{nm}-color.png")
        Image.fromarray(depth).save(root / f"{nm}-depth.png")
        scio.savemat(tool / f"{fi:06d}.mat", {"labels": label, "rois": np.array(rois, dtype=np.float64)})
        frames.append((rgb, depth, label.astype(np.int32), rois))
    sdp = synth.make_state_dict(synth.posenet_spec(K), 21)
    sdr = synth.make_state_dict(synth.refiner_spec(K), 1021)
    torch.save({k: torch.from_numpy(v) for k, v in sdp.items()}, tmp / "pose_model.pth")
    torch.save({k: torch.from_numpy(v) for k, v in sdr.items()}, tmp / "pose_refine_model.pth")
    return root, tool.parent, cfg, frames, sdp, sdr


def test_eval_ycb_entry_point(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_ycb
    from densefusion_amd.lib.preprocess import YCB_CAM, get_bbox
    rng = np.random.default_rng(5)
    K, N, iters = 21, 1000, 2
    root, toolbox, cfg, frames, sdp, sdr = _fabricate(tmp_path, rng, 2, K, N)
    out_wo, out_ref = tmp_path / "wo", tmp_path / "ref"
    eval_ycb.main(["--dataset_root", str(root), "--model", str(tmp_path / "pose_model.pth"), "--refine_model",
                   str(tmp_path / "pose_refine_model.pth"), "--dataset_config_dir", str(cfg), "--ycb_toolbox_dir", str(toolbox),
                   "--result_wo_refine_dir", str(out_wo), "--result_refine_dir", str(out_ref), "--seed", "3"])
    tp, tr = dfnet._to_torch_sd(sdp), dfnet._to_torch_sd(sdr)
    for fi, (rgb, depth, label, rois) in enumerate(frames):
        got_wo = scio.loadmat(out_wo / f"{fi:04d}.mat")["poses"]
        got = scio.loadmat(out_ref / f"{fi:04d}.mat")["poses"]
        assert got.shape == (len(rois), 7) and got_wo.shape == (len(rois), 7)
        for idx, roi in enumerate(rois):
            bb = get_bbox(roi)
            img, cloud, choose, count = preprocess_ref.prepare_object(rgb, depth, label, int(roi[1]), bb, N, 3 + fi * 64 + idx, YCB_CAM)
            with torch.no_grad():
                wo, pose = pose_math.estimate_pose(tp, tr, torch.from_numpy(img)[None], torch.from_numpy(cloud)[None],
                                                   torch.from_numpy(choose), torch.tensor([int(roi[1]) - 1]), iters)
            mp = (rng.random((500, 3)) - 0.5) * 0.2
            add = pose_math.add_metric(pose_math.transform_model(got[idx], mp), pose_math.transform_model(pose, mp))
            add_wo = pose_math.add_metric(pose_math.transform_model(got_wo[idx], mp), pose_math.transform_model(wo, mp))
            assert add < 1e-4 and add_wo < 1e-4, (fi, idx, add, add_wo)


def test_eval_ycb_results_do_not_depend_on_the_window(tmp_path):
    """Detections bucketed by crop size ACROSS a window of frames and run as one device call give, file for file, the
    .mat results of the frame-by-frame path; a degenerate PoseCNN box is written as a lost detection (zero pose,
    tools/eval_ycb.py:234-237) instead of aborting the run."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_ycb
    rng = np.random.default_rng(9)
    K, N = 21, 1000
    root, toolbox, cfg, frames, sdp, sdr = _fabricate(tmp_path, rng, 5, K, N)
    # frame 1 gets a third, degenerate detection (x2 <= x1 + 2)
    meta_path = toolbox / "results_PoseCNN_RSS2018" / "000001.mat"
    meta = scio.loadmat(meta_path)
    rois = np.vstack([meta["rois"], [0, 5, 100, 50, 101, 120, 0.3]])
    scio.savemat(meta_path, {"labels": meta["labels"], "rois": rois})
    outs = {}
    for window in (1, 4):
        wo, ref = tmp_path / f"wo{window}", tmp_path / f"ref{window}"
        eval_ycb.main(["--dataset_root", str(root), "--model", str(tmp_path / "pose_model.pth"), "--refine_model",
                       str(tmp_path / "pose_refine_model.pth"), "--dataset_config_dir", str(cfg), "--ycb_toolbox_dir", str(toolbox),
                       "--result_wo_refine_dir", str(wo), "--result_refine_dir", str(ref), "--seed", "3", "--window", str(window),
                       "--workers", "3"])
        outs[window] = [(scio.loadmat(wo / f"{fi:04d}.mat")["poses"], scio.loadmat(ref / f"{fi:04d}.mat")["poses"]) for fi in range(5)]
    for fi in range(5):
        for a, b in zip(outs[1][fi], outs[4][fi]):
            assert a.shape == b.shape and np.array_equal(a, b), fi
    assert outs[4][1][1].shape == (3, 7) and not outs[4][1][1][2].any() and outs[4][1][1][0].any()
