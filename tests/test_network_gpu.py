"""GPU: PoseNet / PoseRefineNet / the on-device refine loop through the C ABI, against the golden
vectors of the imported reference (tests/golden) and the CPU oracle on the same seeded inputs.

Tolerances: per-tensor max-abs error <= 2e-4 x tensor scale for raw network outputs (fp32 MFMA vs
oneDNN/MKL accumulation order through ~25 un-normalised layers); final poses are compared through
the ADD of the transformed model points, bar 1e-4 m (north_star)."""
import os

import numpy as np
import pytest
import torch

from densefusion_amd import synth
from oracle import dfnet, pose_math

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CAMS = {"cfg1_linemod_80": synth.LINEMOD_CAM, "cfg2_linemod_120x160": synth.LINEMOD_CAM}
ADD_TOL = 1e-4


def _nets(K, N, wseed):
    from densefusion_amd.lib.network import PoseNet, PoseRefineNet
    est, ref = PoseNet(N, K), PoseRefineNet(N, K)
    est.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), wseed).items()}, strict=True)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K), wseed + 1000).items()}, strict=True)
    return est.cuda().eval(), ref.cuda().eval()


def _case(name):
    g = np.load(os.path.join(G, name + ".npz"))
    K, N, H, W, iters, wseed, iseed = [int(v) for v in g["meta"]]
    o = synth.make_object(iseed, H, W, N, K, cam=CAMS.get(name, synth.YCB_CAM))
    return g, (K, N, H, W, iters, wseed), o


def _close(a, b, rtol=2e-4):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-12)
    err = np.abs(a - b).max()
    assert err <= rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def _add(p, q, mp):
    return pose_math.add_metric(pose_math.transform_model(p, mp), pose_math.transform_model(q, mp))


@pytest.mark.parametrize("name", ["tiny", "cfg1_linemod_80", "cfg2_linemod_120x160", "cfg3_ycb_160", "cfg3_ycb_80x120",
                                  "cfg3_ycb_240x320", "cfg3_ycb_480x640", "cfg5_n2000_240x320"])
def test_posenet_forward_golden(name):
    g, (K, N, H, W, iters, wseed), o = _case(name)
    est, _ = _nets(K, N, wseed)
    T = lambda k: torch.from_numpy(o[k])[None].cuda()
    r, t, c, emb = est(T("img"), T("cloud"), torch.from_numpy(o["choose"]).cuda(), torch.from_numpy(o["obj"]).cuda())
    assert r.shape == (1, N, 4) and t.shape == (1, N, 3) and c.shape == (1, N, 1) and emb.shape == (1, 32, N)
    _close(emb, g["emb"]); _close(r, g["out_rx"]); _close(t, g["out_tx"]); _close(c, g["out_cx"])
    assert int(c.view(-1).argmax()) == int(g["which_max"][0])


@pytest.mark.parametrize("name", ["tiny", "cfg2_linemod_120x160", "cfg3_ycb_160", "cfg3_ycb_80x120", "cfg3_ycb_240x320",
                                  "cfg3_ycb_480x640", "cfg5_n2000_240x320"])
def test_estimate_poses_golden(name):
    """df_estimate_poses == the eval loop of the reference (tools/eval_ycb.py:192-229), ADD <= 1e-4 m."""
    from densefusion_amd.lib.network import PoseEstimator
    g, (K, N, H, W, iters, wseed), o = _case(name)
    est, ref = _nets(K, N, wseed)
    pe = PoseEstimator(est, ref)
    T = lambda k: torch.from_numpy(o[k])[None].cuda()
    wo, pose = pe.estimate(T("img"), T("cloud"), torch.from_numpy(o["choose"]).cuda(), torch.from_numpy(o["obj"]).cuda(), iters)
    wo, pose = wo.cpu().numpy()[0], pose.cpu().numpy()[0]
    assert _add(wo, g["pose_wo_refine"], o["model_points"]) < ADD_TOL
    assert _add(pose, g["poses_refined"][-1], o["model_points"]) < ADD_TOL
    if name == "cfg3_ycb_160":   # the YCB eval setting (2 iterations, eval_ycb.py:47) from the same golden
        _, pose2 = pe.estimate(T("img"), T("cloud"), torch.from_numpy(o["choose"]).cuda(), torch.from_numpy(o["obj"]).cuda(), 2)
        assert _add(pose2.cpu().numpy()[0], g["poses_refined"][1], o["model_points"]) < ADD_TOL


def test_refiner_forward_golden_and_api():
    g, (K, N, H, W, iters, wseed), o = _case("tiny")
    est, ref = _nets(K, N, wseed)
    cloud = torch.from_numpy(o["cloud"])[None]
    my_r, my_t = g["pose_wo_refine"][:4].astype(np.float32), g["pose_wo_refine"][4:].astype(np.float32)
    R = torch.from_numpy(pose_math.quaternion_matrix(my_r)[:3, :3].astype(np.float32)).view(1, 3, 3)
    new_cloud = torch.bmm(cloud - torch.from_numpy(my_t).view(1, 1, 3), R)
    pr, pt = ref(new_cloud.cuda(), torch.from_numpy(g["emb"]).cuda(), torch.from_numpy(o["obj"]).cuda())
    assert pr.shape == (1, 4) and pt.shape == (1, 3)
    _close(pr, g["refine0_rx"]); _close(pt, g["refine0_tx"])


def test_batched_objects_equal_solo_runs_and_oracle():
    """Batch extension: B same-size objects in one call == B solo calls (bit-identical), and every
    object agrees with the CPU oracle (ADD of the selected + refined pose <= 1e-4 m)."""
    from densefusion_amd.lib.network import PoseEstimator
    K, N, H, W, B = 21, 1000, 120, 160, 3
    est, ref = _nets(K, N, 13)
    pe = PoseEstimator(est, ref)
    b = synth.make_batch(77, B, H, W, N, K)
    T = lambda k: torch.from_numpy(b[k]).cuda()
    wo, pose = pe.estimate(T("img"), T("cloud"), T("choose"), T("obj"), 2)
    r, t, c, emb = est(T("img"), T("cloud"), T("choose"), T("obj"))
    sdp = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(K), 13))
    sdr = dfnet._to_torch_sd(synth.make_state_dict(synth.refiner_spec(K), 1013))
    for i in range(B):
        sl = lambda k: torch.from_numpy(b[k][i:i + 1]).cuda()
        wo1, pose1 = pe.estimate(sl("img"), sl("cloud"), sl("choose"), sl("obj"), 2)
        assert torch.equal(wo1[0], wo[i]) and torch.equal(pose1[0], pose[i])
        r1, t1, c1, e1 = est(sl("img"), sl("cloud"), sl("choose"), sl("obj"))
        assert torch.equal(r1[0], r[i]) and torch.equal(c1[0], c[i]) and torch.equal(e1[0], emb[i])
        with torch.no_grad():
            args = tuple(torch.from_numpy(b[k][i:i + 1]) for k in ("img", "cloud", "choose", "obj"))
            o_r, o_t, o_c, o_e = dfnet.posenet_forward(sdp, *args)
            cs = torch.sort(o_c.view(-1))[0]
            owo, opose = pose_math.estimate_pose(sdp, sdr, *args, 2)
        _close(emb[i], o_e[0]); _close(r[i], o_r[0]); _close(c[i], o_c[0])
        if float(cs[-1] - cs[-2]) > 1e-4:       # arg-max is discontinuous: only compare away from ties
            assert _add(pose[i].cpu().numpy(), opose, b["model_points"][i]) < ADD_TOL


def test_state_dict_contract_and_errors():
    from densefusion_amd.lib.network import PoseNet, PoseRefineNet
    est, ref = PoseNet(500, 13), PoseRefineNet(500, 13)
    assert [(k, tuple(v.shape)) for k, v in est.state_dict().items()] == synth.posenet_spec(13)
    assert [(k, tuple(v.shape)) for k, v in ref.state_dict().items()] == synth.refiner_spec(13)
    assert sum(p.numel() for p in est.parameters()) == 21440800
    est.cuda()
    r2 = est(torch.zeros(2, 3, 80, 80).cuda(), torch.zeros(2, 500, 3).cuda(), torch.zeros(2, 1, 500, dtype=torch.long).cuda(),
             torch.zeros(2, 1, dtype=torch.long).cuda())                    # train() mode: differentiable path, B objects per call
    assert r2[0].shape == (2, 500, 4) and r2[0].grad_fn is not None and r2[3].shape == (2, 32, 500) and not r2[3].requires_grad
    est.eval()
    with pytest.raises(RuntimeError):                                       # CPU tensors are refused (no CPU path)
        est(torch.zeros(1, 3, 80, 80), torch.zeros(1, 500, 3), torch.zeros(1, 1, 500, dtype=torch.long), torch.zeros(1, 1, dtype=torch.long))
    with pytest.raises(RuntimeError):                                       # wrong point count
        est(torch.zeros(1, 3, 80, 80).cuda(), torch.zeros(1, 400, 3).cuda(), torch.zeros(1, 1, 400, dtype=torch.long).cuda(),
            torch.zeros(1, 1, dtype=torch.long).cuda())


def test_stress_shapes_num_points_2000_and_wide_batch():
    """BASELINE configs[4] shape (num_points = 2000) and a wide batch through the same engine: results must
    still agree with the CPU oracle, and a batch of 24 must equal 24 solo runs bit for bit."""
    from densefusion_amd.lib.network import PoseEstimator
    K, N, H, W = 21, 2000, 80, 120
    est, ref = _nets(K, N, 17)
    pe = PoseEstimator(est, ref)
    b = synth.make_batch(91, 2, H, W, N, K)
    T = lambda k: torch.from_numpy(b[k]).cuda()
    wo, pose = pe.estimate(T("img"), T("cloud"), T("choose"), T("obj"), 2)
    sdp = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(K), 17))
    sdr = dfnet._to_torch_sd(synth.make_state_dict(synth.refiner_spec(K), 1017))
    for i in range(2):
        args = tuple(torch.from_numpy(b[k][i:i + 1]) for k in ("img", "cloud", "choose", "obj"))
        with torch.no_grad():
            o_c = dfnet.posenet_forward(sdp, *args)[2]
            cs = torch.sort(o_c.view(-1))[0]
            owo, opose = pose_math.estimate_pose(sdp, sdr, *args, 2)
        if float(cs[-1] - cs[-2]) > 1e-4:
            assert _add(pose[i].cpu().numpy(), opose, b["model_points"][i]) < ADD_TOL
    K, N, H, W, B = 13, 500, 80, 80, 24
    est, ref = _nets(K, N, 12)
    pe = PoseEstimator(est, ref)
    b = synth.make_batch(92, B, H, W, N, K, cam=synth.LINEMOD_CAM)
    T = lambda k: torch.from_numpy(b[k]).cuda()
    wo, pose = pe.estimate(T("img"), T("cloud"), T("choose"), T("obj"), 4)
    for i in (0, 7, 23):
        sl = lambda k: torch.from_numpy(b[k][i:i + 1]).cuda()
        _, p1 = pe.estimate(sl("img"), sl("cloud"), sl("choose"), sl("obj"), 4)
        assert torch.equal(p1[0], pose[i])


@pytest.mark.parametrize("poison", [float("nan"), 1e30])
def test_results_do_not_depend_on_workspace_garbage(poison):
    """The engine computes (and ignores) padded point rows and unused pyramid rows; nothing a caller sees may
    depend on what the scratch memory held before: poison the workspace and require bit-identical outputs."""
    from densefusion_amd.lib.network import PoseEstimator
    K, N, H, W = 21, 1000, 80, 120
    est, ref = _nets(K, N, 13)
    pe = PoseEstimator(est, ref)
    b = synth.make_batch(55, 2, H, W, N, K)
    T = lambda k: torch.from_numpy(b[k]).cuda()
    r0 = est(T("img"), T("cloud"), T("choose"), T("obj"))
    p0 = pe.estimate(T("img"), T("cloud"), T("choose"), T("obj"), 2)
    p0 = (p0[0].clone(), p0[1].clone())
    est._ws.view(torch.float32).fill_(poison)
    pe._ws.view(torch.float32).fill_(poison)
    r1 = est(T("img"), T("cloud"), T("choose"), T("obj"))
    p1 = pe.estimate(T("img"), T("cloud"), T("choose"), T("obj"), 2)
    for a, c in zip(r0 + p0, r1 + p1):
        assert torch.equal(a, c)


@pytest.mark.parametrize("H,W", [(100, 140), (88, 72)])
def test_crop_sizes_that_are_not_multiples_of_8(H, W):
    """The reference's feature map is 8*ceil-ish(H/8) wide for such crops and `choose` indexes THAT map
    (lib/network.py:98-102 views it flat); the engine must reproduce the same geometry and values."""
    K, N = 13, 500
    est, ref = _nets(K, N, 12)
    o = synth.make_object(77 + H, H, W, N, K, cam=synth.LINEMOD_CAM)
    T = lambda k: torch.from_numpy(o[k])[None]
    sdp = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(K), 12))
    with torch.no_grad():
        want = dfnet.posenet_forward(sdp, T("img"), T("cloud"), torch.from_numpy(o["choose"]), torch.from_numpy(o["obj"]))
    got = est(T("img").cuda(), T("cloud").cuda(), torch.from_numpy(o["choose"]).cuda(), torch.from_numpy(o["obj"]).cuda())
    for a, b in zip(got, want):
        _close(a, b.numpy())


def test_estimate_selects_the_same_pose_as_the_full_forward():
    """The eval loop evaluates the r / t towers only at the arg-max-confidence point (engine `sel` path); the pose it
    reports without refinement must be the one tools/eval_ycb.py:193-203 would pick from the full forward outputs."""
    import numpy as np
    from densefusion_amd import synth
    from densefusion_amd.lib.network import PoseEstimator, PoseNet, PoseRefineNet
    K, N = 21, 1000
    dev = torch.device("cuda:0")
    est, ref = PoseNet(N, K).to(dev).eval(), PoseRefineNet(N, K).to(dev).eval()
    est.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 5).items()})
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K), 1005).items()})
    batch = synth.make_batch(4321, 6, 120, 160, N, K, 500, cam=synth.YCB_CAM)
    img, cloud = torch.from_numpy(batch["img"]).to(dev), torch.from_numpy(batch["cloud"]).to(dev)
    choose, obj = torch.from_numpy(batch["choose"]).to(dev), torch.from_numpy(batch["obj"]).to(dev)
    pose_wo, _ = PoseEstimator(est, ref).estimate(img, cloud, choose, obj, 0)
    out_r, out_t, out_c, _ = est(img, cloud, choose, obj)
    which = out_c.reshape(6, N).argmax(dim=1)
    ar = torch.arange(6, device=dev)
    q = out_r[ar, which]
    q = q / q.norm(dim=1, keepdim=True)
    t = cloud[ar, which] + out_t[ar, which]
    np.testing.assert_allclose(pose_wo[:, :4].cpu().numpy(), q.double().cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(pose_wo[:, 4:].cpu().numpy(), t.double().cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("H,W,B", [(480, 640, 1), (240, 320, 2), (160, 200, 2), (200, 240, 1)])
def test_large_crops_vs_oracle(H, W, B):
    """The largest crop the reference can produce (480x640, datasets/ycb/dataset.py:247-289) and the bench's large
    buckets -- every Winograd-domain / direct choice of the trunk (dilation 1, 2, 4 on 20x25 ... 60x80 maps) and the
    chosen-pixel up_3 path -- against the CPU oracle: raw network outputs and the refined pose."""
    from densefusion_amd.lib.network import PoseEstimator
    K, N = 21, 1000
    est, ref = _nets(K, N, 29)
    b = synth.make_batch(H * 7 + W, B, H, W, N, K)
    T = lambda k: torch.from_numpy(b[k]).cuda()
    _, pose = PoseEstimator(est, ref).estimate(T("img"), T("cloud"), T("choose"), T("obj"), 2)
    r, t, c, emb = est(T("img"), T("cloud"), T("choose"), T("obj"))
    sdp = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(K), 29))
    sdr = dfnet._to_torch_sd(synth.make_state_dict(synth.refiner_spec(K), 1029))
    for i in range(B):
        with torch.no_grad():
            args = tuple(torch.from_numpy(b[k][i:i + 1]) for k in ("img", "cloud", "choose", "obj"))
            o_r, o_t, o_c, o_e = dfnet.posenet_forward(sdp, *args)
            cs = torch.sort(o_c.view(-1))[0]
            _, opose = pose_math.estimate_pose(sdp, sdr, *args, 2)
        _close(emb[i], o_e[0]); _close(r[i], o_r[0]); _close(t[i], o_t[0]); _close(c[i], o_c[0])
        if float(cs[-1] - cs[-2]) > 1e-4:
            assert _add(pose[i].cpu().numpy(), opose, b["model_points"][i]) < ADD_TOL


def test_multi_bucket_call_equals_per_bucket_calls():
    """df_estimate_poses_multi: a window of detections of different crop sizes in one call (shared launches for everything
    that does not depend on the crop geometry) == one df_estimate_poses call per crop size, bit for bit; includes a crop whose
    dilated layers skip the Winograd route, a non-multiple-of-8 crop and a one-object bucket."""
    from densefusion_amd.lib.network import PoseEstimator
    K, N = 21, 1000
    est, ref = _nets(K, N, 13)
    pe = PoseEstimator(est, ref)
    shapes = [(3, 80, 80), (2, 120, 160), (1, 92, 108), (2, 160, 160)]
    bs = [synth.make_batch(500 + i, B, H, W, N, K) for i, (B, H, W) in enumerate(shapes)]
    T = lambda b, k: torch.from_numpy(b[k]).cuda()
    cat = lambda k: torch.cat([T(b, k) for b in bs])
    for iters in (2, 0):
        wo, pose = pe.estimate_multi([T(b, "img") for b in bs], cat("cloud"), cat("choose").reshape(-1, N), cat("obj").reshape(-1), iters)
        assert wo.shape == (8, 7) and pose.shape == (8, 7)
        o = 0
        for b, (B, H, W) in zip(bs, shapes):
            wo1, pose1 = PoseEstimator(est, ref).estimate(T(b, "img"), T(b, "cloud"), T(b, "choose"), T(b, "obj"), iters)
            assert torch.equal(wo1, wo[o:o + B]) and torch.equal(pose1, pose[o:o + B]), (H, W, iters)
            o += B
    # and the oracle agrees on one object of the odd-sized bucket
    sdp = dfnet._to_torch_sd(synth.make_state_dict(synth.posenet_spec(K), 13))
    sdr = dfnet._to_torch_sd(synth.make_state_dict(synth.refiner_spec(K), 1013))
    with torch.no_grad():
        args = tuple(torch.from_numpy(bs[2][k][0:1]) for k in ("img", "cloud", "choose", "obj"))
        _, opose = pose_math.estimate_pose(sdp, sdr, *args, 2)
    _, pose = pe.estimate_multi([T(b, "img") for b in bs], cat("cloud"), cat("choose").reshape(-1, N), cat("obj").reshape(-1), 2)
    assert _add(pose[5].cpu().numpy(), opose, bs[2]["model_points"][0]) < ADD_TOL
    with pytest.raises(RuntimeError):
        pe.estimate_multi([T(bs[0], "img")], cat("cloud"), cat("choose"), cat("obj"), 2)      # object counts disagree


def test_multi_bucket_forward_equals_per_bucket_forwards():
    """df_posenet_forward_multi (PoseNet.forward_multi): the four heads of the full forward for a window of mixed crop sizes in one pass ==
    one forward per crop size, bit for bit (what the refiner phase of tools/train.py asks of its frozen estimator)."""
    K, N = 21, 1000
    est, _ = _nets(K, N, 13)
    shapes = [(3, 80, 80), (2, 120, 160), (1, 92, 108), (2, 160, 160)]
    bs = [synth.make_batch(700 + i, B, H, W, N, K) for i, (B, H, W) in enumerate(shapes)]
    T = lambda b, k: torch.from_numpy(b[k]).cuda()
    cat = lambda k: torch.cat([T(b, k) for b in bs])
    outs = est.forward_multi([T(b, "img") for b in bs], cat("cloud"), cat("choose"), cat("obj"))
    assert [tuple(o.shape) for o in outs] == [(8, N, 4), (8, N, 3), (8, N, 1), (8, 32, N)]
    o = 0
    for b, (B, H, W) in zip(bs, shapes):
        one = est(T(b, "img"), T(b, "cloud"), T(b, "choose"), T(b, "obj"))
        for a, m in zip(one, outs):
            assert torch.equal(a, m[o:o + B]), (H, W)
        o += B
    with pytest.raises(RuntimeError):
        est.forward_multi([T(bs[0], "img")], cat("cloud"), cat("choose"), cat("obj"))      # object counts disagree
    est.train()
    with pytest.raises(RuntimeError):
        est.forward_multi([T(b, "img") for b in bs], cat("cloud"), cat("choose"), cat("obj"))  # inference-only entry point
    est.eval()


def test_a_window_with_more_than_64_crop_sizes():
    """40-pixel snapping yields up to 12 x 16 crop sizes, so a long evaluation window can hold more than 64 of them (the limit of
    df_estimate_poses_multi until round 3): 70 one-object buckets in one call == the same objects evaluated size by size."""
    from densefusion_amd.lib.network import PoseEstimator
    K, N = 3, 128
    est, ref = _nets(K, N, 17)
    pe = PoseEstimator(est, ref)
    shapes = [(1, H, W) for H in (40, 80, 120, 160, 200) for W in range(40, 40 * 15, 40)]
    assert len(shapes) == 70
    bs = [synth.make_batch(900 + i, B, H, W, N, K) for i, (B, H, W) in enumerate(shapes)]
    T = lambda b, k: torch.from_numpy(b[k]).cuda()
    cat = lambda k: torch.cat([T(b, k) for b in bs])
    wo, pose = pe.estimate_multi([T(b, "img") for b in bs], cat("cloud"), cat("choose").reshape(-1, N), cat("obj").reshape(-1), 2)
    solo = PoseEstimator(est, ref)
    for i in (0, 13, 41, 69):
        wo1, pose1 = solo.estimate(T(bs[i], "img"), T(bs[i], "cloud"), T(bs[i], "choose"), T(bs[i], "obj"), 2)
        assert torch.equal(wo1, wo[i:i + 1]) and torch.equal(pose1, pose[i:i + 1]), shapes[i]


_POINT_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from densefusion_amd import synth
from densefusion_amd.lib.network import PoseEstimator, PoseNet, PoseRefineNet
K, N = 21, 1000
est, ref = PoseNet(N, K), PoseRefineNet(N, K)
est.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.posenet_spec(K), 13).items()})
ref.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.refiner_spec(K), 1013).items()})
est, ref = est.cuda().eval(), ref.cuda().eval()
out = []
for seed, B, H, W in ((700, 3, 80, 120), (701, 2, 160, 160)):
    b = synth.make_batch(seed, B, H, W, N, K)
    T = lambda k: torch.from_numpy(b[k]).cuda()
    out += [t.cpu() for t in est(T("img"), T("cloud"), T("choose"), T("obj"))]                       # PoseNet.forward: r, t, c, emb
    out += [t.cpu() for t in PoseEstimator(est, ref).estimate(T("img"), T("cloud"), T("choose"), T("obj"), 2)]
    out += [t.cpu() for t in ref(T("cloud"), out[-3].cuda(), T("obj"))]                               # stand-alone refiner
torch.save(out, sys.argv[2])
"""


def test_fused_point_layers_are_bit_identical_to_the_layer_by_layer_launches(tmp_path):
    """csrc/pointfeat.hip chains conv1 -> conv2 and e_conv1 -> e_conv2 (K = 3 / 32 / 64) inside one launch with the intermediates
    in LDS; every output element still adds the same products in the same order as the separate GEMM launches, so PoseNet's
    outputs, the refined poses and the stand-alone refiner's outputs must not change by a bit (two child processes: the dev switch
    DF_POINT_UNFUSED is read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for tag, env in (("fused", {}), ("unfused", {"DF_DEV_LIB": "1", "DF_POINT_UNFUSED": "1"})):
        f = str(tmp_path / f"{tag}.pt")
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", _POINT_SCRIPT, root, f], check=True, env=e, timeout=600)
        files.append(f)
    a, b = torch.load(files[0]), torch.load(files[1])
    assert len(a) == len(b) == 16
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), f"output {i}: fused and layer-by-layer forms disagree, max diff {float((u.double() - v.double()).abs().max())}"


def test_layer_taps_match_the_references_intermediates():
    """The engine's debug taps (df_net_debug_taps) against the 10 intermediates the imported reference produced for the tiny
    config (forward hooks in oracle/make_golden.py): a regression localises to a layer.  up_3 exists at the chosen pixels only."""
    g, (K, N, H, W, iters, wseed), o = _case("tiny")
    est, _ = _nets(K, N, wseed)
    est.debug_taps(True)
    T = lambda k: torch.from_numpy(o[k])[None].cuda()
    est(T("img"), T("cloud"), torch.from_numpy(o["choose"]).cuda(), torch.from_numpy(o["obj"]).cuda())
    for name in ("stem", "layer1", "layer2", "layer3", "layer4", "psp", "up_1", "up_2"):
        got = est.debug_tap(name).permute(0, 3, 1, 2)                     # NHWC -> the reference's NCHW
        _close(got, g["tap_" + name])
    _close(est.debug_tap("ap_x").reshape(1, 1024, 1), g["tap_ap_x"])
    up3 = est.debug_tap("up_3")[0, :N, :, 0]                              # [N][64] rows of the chosen pixels
    want = torch.from_numpy(g["tap_up_3"])[0].reshape(64, H * W)[:, torch.from_numpy(o["choose"]).reshape(-1)].T
    _close(up3, want)
    est.debug_taps(False)
    with pytest.raises(RuntimeError):
        est.debug_tap("psp")
