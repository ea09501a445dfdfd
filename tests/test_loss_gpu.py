"""GPU: Loss / Loss_refine / ADD(-S) metric through the C ABI against the reference goldens
(non-symmetric branch, run by the imported reference) and the CPU oracle (symmetric branch)."""
import os

import numpy as np
import pytest
import torch

from densefusion_amd import synth
from oracle import loss_ref, pose_math

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, rtol=2e-5):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-12), (np.abs(a - b).max(), np.abs(b).max())


def test_loss_and_loss_refine_golden_nonsymmetric():
    from densefusion_amd.lib.loss import Loss
    from densefusion_amd.lib.loss_refiner import Loss_refine
    g = np.load(os.path.join(G, "loss_nonsym.npz"))
    T = lambda k: torch.from_numpy(g[k]).cuda()
    M = g["target"].shape[1]
    idx = torch.tensor([[3]]).cuda()
    loss, dis, npts, ntgt = Loss(M, [7, 8])(T("pred_r"), T("pred_t"), T("pred_c"), T("target"), T("model_points"), idx,
                                            T("points"), 0.015, False)
    assert loss.dim() == 0 and dis.dim() == 0 and npts.shape == (1, g["points"].shape[1], 3) and ntgt.shape == (1, M, 3)
    _close(loss, g["loss"]); _close(dis, g["dis"]); _close(npts, g["new_points"]); _close(ntgt, g["new_target"])
    d2, np2, nt2 = Loss_refine(M, [7, 8])(T("r_pred_r"), T("r_pred_t"), ntgt, T("model_points"), idx, npts)
    assert d2.shape == (1,)
    _close(d2, g["r_dis"]); _close(np2, g["r_new_points"]); _close(nt2, g["r_new_target"])


@pytest.mark.parametrize("N,M,refine", [(96, 80, False), (500, 500, False), (64, 2600, False), (200, 500, True)])
def test_loss_symmetric_vs_oracle(N, M, refine):
    """Symmetric objects: fused transform + 1-NN + reduction == oracle (materialise, knn_ref.c, gather)."""
    from densefusion_amd.lib.loss import Loss
    from densefusion_amd.lib.loss_refiner import Loss_refine
    rng = np.random.Generator(np.random.PCG64(N * 3 + M))
    o = synth.make_object(9000 + N, 80, 80, N, 13, num_points_mesh=M)
    q = rng.standard_normal((1, N, 4)).astype(np.float32)
    pt = (rng.standard_normal((1, N, 3)) * 0.03).astype(np.float32)
    pc = rng.uniform(0.05, 0.95, (1, N, 1)).astype(np.float32)
    tgt, mp, pts = o["target"][None], o["model_points"][None], o["cloud"][None]
    idx = torch.tensor([[7]])
    C = lambda a: torch.from_numpy(a)
    want = loss_ref.loss_calculation(C(q), C(pt), C(pc), C(tgt), C(mp), idx, C(pts), 0.015, refine, M, [7, 8])
    got = Loss(M, [7, 8])(C(q).cuda(), C(pt).cuda(), C(pc).cuda(), C(tgt).cuda(), C(mp).cuda(), idx.cuda(), C(pts).cuda(), 0.015, refine)
    for a, b in zip(got, want):
        _close(a, b, 5e-5)
    q1 = rng.standard_normal((1, 4)).astype(np.float32)
    t1 = (rng.standard_normal((1, 3)) * 0.02).astype(np.float32)
    wantr = loss_ref.loss_refine_calculation(C(q1), C(t1), want[3], C(mp), idx, want[2], M, [7, 8])
    gotr = Loss_refine(M, [7, 8])(C(q1).cuda(), C(t1).cuda(), got[3], C(mp).cuda(), idx.cuda(), got[2])
    for a, b in zip(gotr, wantr):
        _close(a, b, 5e-5)


def test_add_metric_vs_oracle_and_ply_fixture():
    from densefusion_amd.lib.metric import add_metric
    rng = np.random.Generator(np.random.PCG64(4))
    B, M = 5, 500
    objs = [synth.make_object(700 + i, 80, 80, 64, 13, num_points_mesh=M) for i in range(B)]
    mp = np.stack([o["model_points"] for o in objs]); tg = np.stack([o["target"] for o in objs])
    pose = np.zeros((B, 7))
    for i in range(B):
        pose[i, :4] = synth.random_unit_quaternion(rng)
        pose[i, 4:] = objs[i]["cloud"].mean(0) + rng.standard_normal(3) * 0.01
    sym = np.array([0, 1, 0, 1, 1], dtype=np.int32)
    got = add_metric(torch.from_numpy(pose).cuda(), torch.from_numpy(mp).cuda(), torch.from_numpy(tg).cuda(), sym).cpu().numpy()
    for i in range(B):
        pred = pose_math.transform_model(pose[i], mp[i])
        want = pose_math.adds_metric(pred, tg[i]) if sym[i] else pose_math.add_metric(pred, tg[i])
        assert abs(got[i] - want) <= 2e-6 * max(1.0, want), (i, got[i], want)
    # the two clouds the reference ships: identity pose on `pred` as the model
    g = np.load(os.path.join(G, "ply_clouds.npz"))
    ident = torch.tensor([[1.0, 0, 0, 0, 0, 0, 0]], dtype=torch.float64).cuda()
    P, Tt = torch.from_numpy(g["pred"].astype(np.float32))[None].cuda(), torch.from_numpy(g["target"].astype(np.float32))[None].cuda()
    assert abs(add_metric(ident, P, Tt)[0].item() - 0.0168566) < 2e-6
    assert abs(add_metric(ident, P, Tt, [1])[0].item() - 0.0092865) < 2e-6
    assert abs(add_metric(ident, Tt, P, [1])[0].item() - 0.0095001) < 2e-6


def test_loss_argument_errors():
    from densefusion_amd.lib.loss import Loss
    z = lambda *s: torch.zeros(*s).cuda()
    with pytest.raises(RuntimeError):
        Loss(500, [])(z(1, 10, 4), z(1, 10, 3), z(1, 10, 1), z(1, 400, 3), z(1, 500, 3), torch.tensor([[0]]).cuda(), z(1, 10, 3), 0.015, False)
    with pytest.raises(RuntimeError):
        Loss(500, [])(torch.zeros(1, 10, 4), z(1, 10, 3), z(1, 10, 1), z(1, 500, 3), z(1, 500, 3), torch.tensor([[0]]), z(1, 10, 3), 0.015, False)


@pytest.mark.parametrize("sym", [False, True])
def test_loss_backward_vs_autograd_of_oracle(sym):
    """loss.backward() / dis.backward() through the fused kernels == torch autograd through the CPU restatement."""
    from densefusion_amd.lib.loss import Loss
    from densefusion_amd.lib.loss_refiner import Loss_refine
    rng = np.random.Generator(np.random.PCG64(42 + sym))
    N, M = 120, 90
    o = synth.make_object(9100, 80, 80, N, 13, num_points_mesh=M)
    q = rng.standard_normal((1, N, 4)).astype(np.float32)
    pt = (rng.standard_normal((1, N, 3)) * 0.03).astype(np.float32)
    pc = rng.uniform(0.05, 0.95, (1, N, 1)).astype(np.float32)
    tgt, mp, pts = o["target"][None], o["model_points"][None], o["cloud"][None]
    idx = torch.tensor([[7 if sym else 3]])
    C = lambda a: torch.from_numpy(a)
    # CPU: autograd through the oracle
    cq, ct, cc = C(q).clone().requires_grad_(), C(pt).clone().requires_grad_(), C(pc).clone().requires_grad_()
    want = loss_ref.loss_calculation(cq, ct, cc, C(tgt), C(mp), idx, C(pts), 0.015, False, M, [7, 8])
    want[0].backward()
    # GPU
    gq, gt, gc = C(q).cuda().requires_grad_(), C(pt).cuda().requires_grad_(), C(pc).cuda().requires_grad_()
    got = Loss(M, [7, 8])(gq, gt, gc, C(tgt).cuda(), C(mp).cuda(), idx.cuda(), C(pts).cuda(), 0.015, False)
    assert got[0].requires_grad and not got[2].requires_grad
    got[0].backward()
    _close(got[0], want[0], 5e-5)
    for a, b in ((gq.grad, cq.grad), (gt.grad, ct.grad), (gc.grad, cc.grad)):
        _close(a, b, 2e-4)
    # refiner loss
    q1 = rng.standard_normal((1, 4)).astype(np.float32)
    t1 = (rng.standard_normal((1, 3)) * 0.02).astype(np.float32)
    cq1, ct1 = C(q1).clone().requires_grad_(), C(t1).clone().requires_grad_()
    wr = loss_ref.loss_refine_calculation(cq1, ct1, want[3], C(mp), idx, want[2], M, [7, 8])
    wr[0].backward()
    gq1, gt1 = C(q1).cuda().requires_grad_(), C(t1).cuda().requires_grad_()
    gr = Loss_refine(M, [7, 8])(gq1, gt1, got[3], C(mp).cuda(), idx.cuda(), got[2])
    gr[0].backward()
    _close(gr[0], wr[0], 5e-5)
    _close(gq1.grad, cq1.grad, 2e-4); _close(gt1.grad, ct1.grad, 2e-4)


def test_symmetric_loss_matches_are_df_knn_bit_for_bit():
    """BASELINE configs[3] size (N = 1000 per-point poses, M = 500 mesh points: 250 M pairs): the nearest-neighbour choice
    inside the fused loss kernel (csrc/loss.hip add_dis_sym_kernel) == KNearestNeighbor(1)(target, pred) on the materialised
    transformed points (lib/loss.py:41-47 as intended), index for index; both run the scan of csrc/knn_core.h."""
    import ctypes
    from densefusion_amd import _lib
    from densefusion_amd.lib.knn import KNearestNeighbor
    N, M = 1000, 500
    rng = np.random.Generator(np.random.PCG64(31))
    o = synth.make_object(9300, 120, 160, N, 21, num_points_mesh=M)
    q = torch.from_numpy(rng.standard_normal((N, 4)).astype(np.float32)).cuda()
    pt = torch.from_numpy((rng.standard_normal((N, 3)) * 0.03).astype(np.float32)).cuda()
    pc = torch.from_numpy(rng.uniform(0.05, 0.95, N).astype(np.float32)).cuda()
    tgt, mp, pts = (torch.from_numpy(o[k]).cuda() for k in ("target", "model_points", "cloud"))
    loss, dis = torch.empty(1).cuda(), torch.empty(1).cuda()
    npts, ntgt, scratch = torch.empty(N, 3).cuda(), torch.empty(M, 3).cuda(), torch.empty(N).cuda()
    sel = torch.empty(N, M, dtype=torch.int32).cuda()
    _lib.check(_lib.lib().df_loss_forward(q.data_ptr(), pt.data_ptr(), pc.data_ptr(), tgt.data_ptr(), mp.data_ptr(), pts.data_ptr(), N, M,
                                          ctypes.c_float(0.015), 1, loss.data_ptr(), dis.data_ptr(), npts.data_ptr(), ntgt.data_ptr(),
                                          scratch.data_ptr(), sel.data_ptr(), _lib.current_stream()), "loss_forward")
    # materialise pred exactly as the kernel forms it (fp32, the reference's expression order: lib/loss.py:16-38)
    qn = q / torch.sqrt(((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3])[:, None]
    a, b, c, d = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    R = torch.stack([1.0 - 2.0 * (c * c + d * d), 2.0 * b * c - 2.0 * a * d, 2.0 * a * c + 2.0 * b * d,
                     2.0 * b * c + 2.0 * d * a, 1.0 - 2.0 * (b * b + d * d), -2.0 * a * b + 2.0 * c * d,
                     -2.0 * a * c + 2.0 * b * d, 2.0 * a * b + 2.0 * c * d, 1.0 - 2.0 * (b * b + c * c)], dim=1).view(N, 3, 3)
    t = pts + pt
    x, y, z = mp[:, 0][None], mp[:, 1][None], mp[:, 2][None]
    pred = torch.stack([(x * R[:, 0, 0:1] + y * R[:, 0, 1:2] + z * R[:, 0, 2:3]) + t[:, 0:1],
                        (x * R[:, 1, 0:1] + y * R[:, 1, 1:2] + z * R[:, 1, 2:3]) + t[:, 1:2],
                        (x * R[:, 2, 0:1] + y * R[:, 2, 1:2] + z * R[:, 2, 2:3]) + t[:, 2:3]], dim=0)      # [3][N][M]
    inds = KNearestNeighbor(1)(tgt.t().contiguous()[None], pred.reshape(1, 3, N * M).contiguous())
    want = (inds.view(N, M) - 1).to(torch.int32)
    mism = int((want != sel).sum())
    # the quaternion normalisation goes through sqrt / division on both sides; any pred point that differs in its last bit
    # could flip a near-tie, so tolerate none only where the materialised points are bit-equal to the kernel's: compare via dis
    assert mism <= N * M // 100000, f"{mism} of {N * M} matches differ"
    sel_l = sel.long()
    gathered = tgt[sel_l.reshape(-1)].view(N, M, 3)
    dis_ref = torch.norm(pred.permute(1, 2, 0) - gathered, dim=2).mean(1)
    _close(scratch, dis_ref, 2e-6)


@pytest.mark.parametrize("refine", [False, True])
def test_stacked_frames_give_each_frames_own_numbers(refine):
    """df_loss_forward_frames (Loss.forward_frames): B stacked frames, symmetric and not, in a handful of launches == B Loss.forward calls, bit for
    bit (the trainer's passes and the refiner phase's re-centring call it instead of looping over the frames)."""
    from densefusion_amd.lib.loss import Loss
    B, N, M = 5, 200, 120
    rng = np.random.Generator(np.random.PCG64(77))
    objs = [synth.make_object(9100 + b, 80, 80, N, 13, num_points_mesh=M) for b in range(B)]
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    q, pt = C(rng.standard_normal((B, N, 4)).astype(np.float32)), C((rng.standard_normal((B, N, 3)) * 0.03).astype(np.float32))
    pc = C(rng.uniform(0.05, 0.95, (B, N, 1)).astype(np.float32))
    tgt, mp, pts = (C(np.stack([o[k] for o in objs])) for k in ("target", "model_points", "cloud"))
    ids = [7, 3, 8, 7, 1]                                           # 7 and 8 are symmetric
    crit = Loss(M, [7, 8])
    loss, dis, npts, ntg = crit.forward_frames(q, pt, pc, tgt, mp, ids, pts, 0.015, refine)
    assert loss.shape == (B,) and dis.shape == (B,) and npts.shape == (B, N, 3) and ntg.shape == (B, M, 3)
    for b in range(B):
        one = crit(q[b:b + 1], pt[b:b + 1], pc[b:b + 1], tgt[b:b + 1], mp[b:b + 1], torch.tensor([[ids[b]]]), pts[b:b + 1], 0.015, refine)
        assert torch.equal(one[0].reshape(()), loss[b]) and torch.equal(one[1].reshape(()), dis[b])
        assert torch.equal(one[2][0], npts[b]) and torch.equal(one[3][0], ntg[b])
    with pytest.raises(RuntimeError):
        crit.forward_frames(q, pt, pc, tgt, mp, ids[:3], pts, 0.015, refine)


def test_more_stacked_frames_than_one_launch_table_holds():
    """65 frames: the frame table of a launch holds 60, the rest goes to a second round of launches -- same numbers frame by frame."""
    from densefusion_amd.lib.loss import Loss
    B, N, M = 65, 48, 40
    rng = np.random.Generator(np.random.PCG64(78))
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    q, pt = C(rng.standard_normal((B, N, 4)).astype(np.float32)), C((rng.standard_normal((B, N, 3)) * 0.03).astype(np.float32))
    pc = C(rng.uniform(0.05, 0.95, (B, N, 1)).astype(np.float32))
    tgt, mp = C(rng.standard_normal((B, M, 3)).astype(np.float32) * 0.1), C(rng.standard_normal((B, M, 3)).astype(np.float32) * 0.1)
    pts = C(rng.standard_normal((B, N, 3)).astype(np.float32) * 0.1)
    ids = [7 if b % 3 == 0 else 2 for b in range(B)]
    crit = Loss(M, [7])
    loss, dis, npts, ntg = crit.forward_frames(q, pt, pc, tgt, mp, ids, pts, 0.015, False)
    for b in (0, 1, 59, 60, 63, 64):
        one = crit(q[b:b + 1], pt[b:b + 1], pc[b:b + 1], tgt[b:b + 1], mp[b:b + 1], torch.tensor([[ids[b]]]), pts[b:b + 1], 0.015, False)
        assert torch.equal(one[0].reshape(()), loss[b]) and torch.equal(one[1].reshape(()), dis[b]) and torch.equal(one[2][0], npts[b]), b
