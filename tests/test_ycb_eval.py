"""YCB-Video evaluation: host AUC logic on CPU, device distances on GPU, both vs the numpy oracle."""
import numpy as np
import pytest
import torch

from densefusion_amd import synth
from oracle import ycb_metric


def _auc_cases():
    rng = np.random.default_rng(0)
    yield rng.uniform(0, 0.15, 500)
    yield np.array([0.2, 0.3, np.inf])                 # nothing under the threshold
    yield np.array([0.01, 0.01, 0.05, 0.05, 0.12])      # repeated distances
    yield np.array([0.0])


def test_auc_matches_oracle_and_known_values():
    from densefusion_amd.lib import ycb_eval
    for d in _auc_cases():
        a, c = ycb_eval.auc_and_lt2cm(d)
        ao, co = ycb_metric.auc_and_lt2cm(d)
        assert abs(a - ao) < 1e-12 and abs(c - co) < 1e-12
    # all poses perfect -> AUC 1, all beyond 10 cm -> 0; uniform distances on [0, 0.1] -> 1/2
    assert abs(ycb_eval.auc_and_lt2cm(np.zeros(10))[0] - 1.0) < 1e-12
    assert ycb_eval.auc_and_lt2cm(np.full(10, 0.5))[0] == 0.0
    assert abs(ycb_eval.auc_and_lt2cm(np.linspace(0, 0.1, 100001))[0] - 0.5) < 1e-4


def test_pose_to_rt():
    from densefusion_amd.lib import ycb_eval
    rt = ycb_eval.pose_to_rt([1, 0, 0, 0, 0.1, 0.2, 0.3])
    assert np.allclose(rt, np.hstack([np.eye(3), [[0.1], [0.2], [0.3]]]))


@pytest.mark.gpu
def test_ycb_distances_vs_oracle():
    from densefusion_amd.lib import ycb_eval
    rng = np.random.default_rng(2)
    B, M = 4, 2620                                        # YCB models/*/points.xyz hold 2620 points
    pts = (rng.random((B, M, 3)) - 0.5) * 0.2
    est, gt = np.zeros((B, 3, 4)), np.zeros((B, 3, 4))
    for b in range(B):
        qg = synth.random_unit_quaternion(rng)
        qe = qg + rng.standard_normal(4) * 0.02
        gt[b] = ycb_eval.pose_to_rt(np.concatenate([qg, rng.standard_normal(3) * 0.3]))
        est[b] = ycb_eval.pose_to_rt(np.concatenate([qe / np.linalg.norm(qe), gt[b][:, 3] + rng.standard_normal(3) * 0.005]))
    add, adi = ycb_eval.ycb_distances(torch.from_numpy(est).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(pts).cuda())
    for b in range(B):
        assert abs(add[b].item() - ycb_metric.add(est[b], gt[b], pts[b].T)) < 1e-12
        assert abs(adi[b].item() - ycb_metric.adi(est[b], gt[b], pts[b].T)) < 1e-12
    assert (adi <= add + 1e-15).all()
