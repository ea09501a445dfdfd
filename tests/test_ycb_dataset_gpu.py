"""GPU: the YCB-Video ``PoseDataset`` mirror (real frames, no augmentation) on a fabricated dataset tree against the numpy
restatement of datasets/ycb/dataset.py:90-217 (oracle/ycb_dataset_ref.py)."""
import os
import random

import numpy as np
import pytest
import scipy.io as scio
import torch
from PIL import Image

from densefusion_amd import synth
from oracle import ycb_dataset_ref

pytestmark = pytest.mark.gpu
from fabricate import CLASSES, make_ycb_tree as make_tree  # noqa: E402,F401


@pytest.mark.parametrize("refine", [False, True])
def test_ycb_dataset_matches_restatement(tmp_path, refine):
    from densefusion_amd.datasets.ycb.dataset import PoseDataset
    rng = np.random.default_rng(3)
    root, cfg = str(tmp_path / "YCB"), str(tmp_path / "cfg")
    names = make_tree(root, cfg, rng)
    N = 1000
    ds = PoseDataset("test", N, False, root, 0.0, refine, dataset_config_dir=cfg, seed=5)
    assert len(ds) == 6 and ds.get_sym_list() == [12, 15, 18, 19, 20] and ds.get_num_points_mesh() == (2600 if refine else 500)
    for i, name in enumerate(names):
        np.random.seed(100 + i); random.seed(200 + i)
        got = ds[i]
        np.random.seed(100 + i); random.seed(200 + i)
        rgb = np.array(Image.open(f"{root}/{name}-color.png")); depth = np.array(Image.open(f"{root}/{name}-depth.png"))
        label = np.array(Image.open(f"{root}/{name}-label.png")); meta = scio.loadmat(f"{root}/{name}-meta.mat")
        obj = meta["cls_indexes"].flatten().astype(np.int32)
        while True:                                           # the reference's selection loop (dataset.py:139-146)
            idx = np.random.randint(0, len(obj))
            if np.count_nonzero((label == obj[idx]) & (depth != 0)) > 50:
                break
        for _ in range(3):
            random.uniform(0.0, 0.0)
        cld = ds.cld[int(obj[idx])]
        keep_n = 2600 if refine else 500
        drop = set(random.sample(range(len(cld)), len(cld) - keep_n))
        keep = [j for j in range(len(cld)) if j not in drop]
        cloud, choose, img, target, model_points, box = ycb_dataset_ref.get_item(rgb, depth, label, meta, int(name[5:9]), idx, cld, keep, N,
                                                                                  (5 * 1000003 + i) & 0xFFFFFFFF)
        assert int(got[5][0]) == int(obj[idx]) - 1 and not (i == 1 and obj[idx] == 4)
        assert tuple(got[2].shape[1:]) == (box[1] - box[0], box[3] - box[2])
        assert torch.equal(got[1].cpu(), torch.from_numpy(choose))
        assert torch.equal(got[0].cpu(), torch.from_numpy(cloud))
        assert torch.equal(got[2].cpu(), torch.from_numpy(img))
        np.testing.assert_array_equal(got[4].cpu().numpy(), model_points)
        np.testing.assert_allclose(got[3].cpu().numpy(), target, rtol=0, atol=1e-7)


def test_ycb_training_augmentation(tmp_path):
    """``add_noise=True`` and synthetic frames (datasets/ycb/dataset.py:117-136,149-167,196-221): one translation goes into cloud AND
    target, occluders only ever remove mask pixels, a synthetic frame is pasted over a real background and gets pixel noise."""
    from densefusion_amd.datasets.ycb.dataset import PoseDataset
    rng = np.random.default_rng(4)
    root, cfg = str(tmp_path / "YCB"), str(tmp_path / "cfg")
    make_tree(root, cfg, rng)
    N, nt = 1000, 0.03
    clean = PoseDataset("train", N, False, root, 0.0, False, dataset_config_dir=cfg, seed=5)
    noisy = PoseDataset("train", N, True, root, nt, False, dataset_config_dir=cfg, seed=5)
    assert len(noisy) == 6 and len(noisy.syn) == 2 and len(noisy.real) == 4
    occluded = 0
    for i in range(4):                                            # real frames
        np.random.seed(50 + i); random.seed(60 + i)
        hc = clean.host_item(i)
        np.random.seed(50 + i); random.seed(60 + i)
        hn = noisy.host_item(i)
        assert hn[0].shape == hc[0].shape == (480, 640, 3) and not torch.equal(hn[0], hc[0])         # jittered (and maybe occluded) colours
        lab_c, lab_n = hc[2].numpy(), hn[2].numpy()
        assert ((lab_n == lab_c) | (lab_n == 0)).all()                                                 # occluders only clear labels
        occluded += int((lab_n != lab_c).any())
        add_t = hn[4][5:8].numpy()
        assert (np.abs(add_t) <= nt).all() and np.abs(add_t).max() > 0 and not hc[4][5:8].any()
        # target = model_points R^T + t + add_t for the pose of the object that was drawn
        meta = scio.loadmat(f"{root}/{noisy.list[i]}-meta.mat")
        obj = meta["cls_indexes"].flatten().tolist()
        pose = meta["poses"][:, :, obj.index(int(hn[3][4]))]
        want = hn[6].numpy().astype(np.float64) @ pose[:, :3].T + pose[:, 3] + add_t
        np.testing.assert_allclose(hn[5].numpy(), want, atol=1e-6)
        item = noisy.device_item(i, hn)
        base = noisy.device_item(i, hn[:4] + (torch.cat([hn[4][:5], torch.zeros(3, dtype=torch.float64)]),) + hn[5:])
        np.testing.assert_allclose((item[0] - base[0]).cpu().numpy(), np.broadcast_to(add_t.astype(np.float32), (N, 3)), atol=1e-6)
        assert torch.equal(item[1], base[1]) and torch.equal(item[2], base[2])
    assert occluded >= 1
    for i in (4, 5):                                              # synthetic frames, also without add_noise
        for ds in (clean, noisy):
            np.random.seed(70 + i); random.seed(80 + i)
            h = ds.host_item(i)
            assert int(h[3][5]) == 1
            raw = np.array(Image.open(f"{root}/{ds.list[i]}-color.png"))[:, :, :3]
            lab = np.array(Image.open(f"{root}/{ds.list[i]}-label.png"))
            back = (lab == 0)
            assert (h[0].numpy()[back] != raw[back]).any()                                           # a real frame shows through the background
            a, b = ds.device_item(i, h), ds.device_item(i, h)
            assert a[2].shape == b[2].shape and not torch.equal(a[2], b[2])                            # N(0, 7) pixel noise, drawn per fetch
            assert float((a[2] - b[2]).std()) == pytest.approx(7.0 / 0.226 * 2 ** 0.5, rel=0.1)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    got = noisy[1]                                                # the plain __getitem__ goes the same way
    assert len(got) == 6 and got[0].shape == (N, 3)
