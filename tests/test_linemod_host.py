"""CPU: host helpers of the LineMOD loader against the numpy restatement (oracle/linemod_ref.py) -- box snapping,
blob rectangle, .ply reader, list parsing.  No device call."""
import numpy as np

from densefusion_amd.datasets.linemod import dataset as D
from oracle import linemod_ref


def test_get_bbox_matches_restatement():
    rng = np.random.default_rng(0)
    for _ in range(5000):
        b = [int(rng.integers(-30, 670)), int(rng.integers(-30, 510)), int(rng.integers(0, 660)), int(rng.integers(0, 500))]
        got = D.get_bbox(list(b))
        assert tuple(got) == tuple(linemod_ref.get_bbox(list(b))), b
        rmin, rmax, cmin, cmax = got
        if 0 <= b[0] < 600 and 0 <= b[1] < 440 and 0 < b[2] < 600 and 0 < b[3] < 440:      # a box that starts inside the frame
            assert 0 <= rmin < rmax <= 480 and 0 <= cmin < cmax <= 640


def test_mask_to_bbox():
    rng = np.random.default_rng(1)
    for _ in range(20):
        m = np.zeros((480, 640), dtype=bool)
        for _ in range(int(rng.integers(0, 5))):
            r0, c0 = int(rng.integers(0, 400)), int(rng.integers(0, 560))
            m[r0:r0 + int(rng.integers(1, 80)), c0:c0 + int(rng.integers(1, 80))] |= rng.random((1, 1)) < 2
        assert D.mask_to_bbox(m) == linemod_ref.mask_to_bbox(m)
    assert D.mask_to_bbox(np.zeros((480, 640), dtype=bool)) == [0, 0, 0, 0]
    m = np.zeros((480, 640), dtype=bool)
    m[10, 10] = m[11, 11] = True                 # diagonal neighbours are one blob (8-connectivity, like a traced contour)
    assert D.mask_to_bbox(m) == [10, 10, 2, 2]


def test_ply_reader_and_lists(tmp_path):
    from test_linemod_dataset_gpu import make_tree
    root = make_tree(str(tmp_path / "lm"), frames_per_obj=11)
    a = D.ply_vtx(f"{root}/models/obj_05.ply")
    b = linemod_ref.ply_vtx(f"{root}/models/obj_05.ply")
    assert a.dtype == np.float32 and a.shape == (640, 3) and np.array_equal(a, b)
    ds = D.PoseDataset.__new__(D.PoseDataset)      # list parsing only: no device needed
    D.PoseDataset.__init__(ds, "test", 500, False, root, 0.0, True, device="cpu")
    assert len(ds) == 13 and ds.list_rank[:2] == [27, 27]           # the 10th line of every test.txt
    D.PoseDataset.__init__(ds, "eval", 500, False, root, 0.0, True, device="cpu")
    assert len(ds) == 13 * 11 and ds.list_label[0].endswith("segnet_results/01_label/0000_label.png")
    assert ds._meta(2, 0)["obj_id"] == 2


def test_ycb_dataset_get_bbox_matches_restatement():
    from densefusion_amd.datasets.ycb import dataset as Y
    from oracle import ycb_dataset_ref
    rng = np.random.default_rng(4)
    for _ in range(300):
        m = np.zeros((480, 640), dtype=bool)
        h, w = int(rng.integers(1, 480)), int(rng.integers(1, 640))
        r0, c0 = int(rng.integers(0, 481 - h)), int(rng.integers(0, 641 - w))
        m[r0:r0 + h, c0:c0 + w] = rng.random((h, w)) < 0.5
        m[r0, c0] = m[r0 + h - 1, c0 + w - 1] = True
        got = Y.get_bbox(m)
        assert got == ycb_dataset_ref.get_bbox(m)
        assert 0 <= got[0] < got[1] <= 480 and 0 <= got[2] < got[3] <= 640
