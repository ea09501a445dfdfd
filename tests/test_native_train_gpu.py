"""GPU: the native training step (df_posenet_train_step / df_refiner_train_step, csrc/train.hip) -- forward + loss + backward
of B frames in ONE library call -- against torch CPU autograd through the oracle restatement, against the imported
reference's own backward (tests/golden/grad_tiny.npz), and for the properties the flat-buffer design promises: gradients
accumulate over frames like separate bs = 1 passes, two identical steps give bit-identical gradient buffers, the state dict
round-trips through the kernel layout, the step replays from a hipGraph."""
import os

import numpy as np
import pytest
import torch

from densefusion_amd import synth
from oracle import dfnet, loss_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol, name=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


def _trainer(kind, N, K, sd):
    from densefusion_amd.native_train import NativeTrainer
    tr = NativeTrainer(kind, N, K, DEV)
    tr.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return tr


def _frames(objs, keys=("img", "cloud", "choose", "obj", "target", "model_points")):
    return {k: torch.stack([torch.from_numpy(o[k]) for o in objs]).to(DEV) for k in keys}


def test_state_dict_round_trips_through_the_kernel_layout():
    K, N = 3, 64
    for kind, spec, seed in (("posenet", synth.posenet_spec(K), 5), ("refiner", synth.refiner_spec(K), 6)):
        sd = synth.make_state_dict(spec, seed)
        tr = _trainer(kind, N, K, sd)
        back = tr.state_dict()
        assert list(back) == list(sd)                      # the reference's keys, in its order
        for k, v in sd.items():
            assert tuple(back[k].shape) == v.shape and torch.equal(back[k].cpu(), torch.from_numpy(v)), k
    with pytest.raises(RuntimeError):
        tr.load_state_dict({"feat.conv1.weight": torch.zeros(64, 3, 1)})       # strict: missing keys


@pytest.mark.parametrize("sym", [False, True])
def test_posenet_step_matches_oracle_autograd(sym):
    K, N, H, W, M = 2, 64, 40, 40, 60
    sd = synth.make_state_dict(synth.posenet_spec(K), 11)
    o = synth.make_object(101, H, W, N, K, num_points_mesh=M)
    o["obj"][0] = 1 if sym else 0
    idx = torch.tensor([[int(o["obj"][0])]])
    T = lambda k: torch.from_numpy(o[k])[None]
    psd = {k: torch.from_numpy(v).clone().requires_grad_() for k, v in sd.items()}
    r, t, c, emb = dfnet.posenet_forward(psd, T("img"), T("cloud"), torch.from_numpy(o["choose"]), idx)
    want_loss, want_dis, want_np, want_nt = loss_ref.loss_calculation(r, t, c, T("target"), T("model_points"), idx, T("cloud"), 0.015, False, M, [1])
    want_loss.backward()
    tr = _trainer("posenet", N, K, sd)
    f = _frames([o])
    out = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [sym], 0.015, dropout=False, want_pred=True)
    _close(out["pred_r"], r, 2e-4, "pred_r"); _close(out["pred_c"], c, 2e-4, "pred_c"); _close(out["emb"], emb, 2e-4, "emb")
    _close(out["loss"], want_loss.reshape(1), 1e-4, "loss"); _close(out["dis"], want_dis.reshape(1), 1e-4, "dis")
    _close(out["new_points"], want_np, 1e-4, "new_points"); _close(out["new_target"], want_nt, 1e-4, "new_target")
    got = tr.grad_dict()
    checked = 0
    for key, g in got.items():
        want = psd[key].grad
        if "classifier" in key:
            assert float(g.abs().max()) == 0.0                 # dead weights get no gradient
            continue
        assert want is not None, key
        _close(g, want, 2e-3, key)
        checked += 1
    assert checked >= 70


def test_refiner_step_matches_oracle_autograd():
    K, N, M = 2, 64, 60
    sd = synth.make_state_dict(synth.refiner_spec(K), 1011)
    o = synth.make_object(103, 40, 40, N, K, num_points_mesh=M)
    o["obj"][0] = 1
    emb = torch.from_numpy(np.random.default_rng(0).standard_normal((1, 32, N)).astype(np.float32))
    idx = torch.tensor([[1]])
    T = lambda k: torch.from_numpy(o[k])[None]
    psd = {k: torch.from_numpy(v).clone().requires_grad_() for k, v in sd.items()}
    pr, pt = dfnet.refiner_forward(psd, T("cloud"), emb, idx)
    want, want_np, want_nt = loss_ref.loss_refine_calculation(pr, pt, T("target"), T("model_points"), idx, T("cloud"), M, [1])
    want.backward()
    tr = _trainer("refiner", N, K, sd)
    out = tr.step_refiner(T("cloud").to(DEV), emb.to(DEV), idx.to(DEV), T("target").to(DEV), T("model_points").to(DEV), [True])
    _close(out["dis"], want.reshape(1), 1e-4, "dis"); _close(out["new_points"], want_np, 1e-4, "new_points")
    _close(out["new_target"], want_nt, 1e-4, "new_target")
    for key, g in tr.grad_dict().items():
        _close(g, psd[key].grad, 2e-3, key)


def test_native_step_matches_the_references_backward():
    """tests/golden/grad_tiny.npz: the imported reference's Loss(...).backward() through its PoseNet (oracle/make_golden.py)."""
    from oracle.make_golden import grad_sample
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "grad_tiny.npz"))
    K, N, H, W, M, wseed, iseed, idx0 = [int(v) for v in g["meta"]]
    o = synth.make_object(iseed, H, W, N, K, num_points_mesh=M)
    o["obj"][0] = idx0
    tr = _trainer("posenet", N, K, synth.make_state_dict(synth.posenet_spec(K), wseed))
    f = _frames([o])
    out = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [False], 0.015, dropout=False, want_pred=True)
    _close(out["pred_r"], torch.from_numpy(g["out_rx"]), 2e-4, "out_rx"); _close(out["pred_c"], torch.from_numpy(g["out_cx"]), 2e-4, "out_cx")
    _close(out["loss"], torch.from_numpy(g["loss"]).reshape(1), 1e-4, "loss"); _close(out["dis"], torch.from_numpy(g["dis"]).reshape(1), 1e-4, "dis")
    grads = tr.grad_dict()
    keys = [k[5:] for k in g.files if k.startswith("grad:")]
    assert len(keys) == 20
    for k in keys:
        _close(torch.from_numpy(grad_sample(grads[k].cpu().numpy())), torch.from_numpy(g["grad:" + k]), 2e-3, k)


def test_batched_step_accumulates_like_separate_frames_and_is_bit_reproducible():
    """B frames in one step == the same frames one per step (gradients accumulate in .grad, tools/train.py:161-169), with the
    Dropout2d masks on (same seed -> same masks per frame index is NOT promised across batchings, so dropout is off for that
    comparison); two identical steps leave bit-identical gradient buffers, dropout on, symmetric and plain frames mixed."""
    K, N, H, W, M, B = 3, 128, 40, 80, 60, 3
    sd = synth.make_state_dict(synth.posenet_spec(K), 17)
    objs = [synth.make_object(300 + i, H, W, N, K, num_points_mesh=M) for i in range(B)]
    for i, o in enumerate(objs):
        o["obj"][0] = i % K
    sym = [int(o["obj"][0]) == 1 for o in objs]
    f = _frames(objs)
    tr = _trainer("posenet", N, K, sd)
    a = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=False)
    g_batched = tr.grad.clone()
    tr.zero_grad()
    for b in range(B):
        s = slice(b, b + 1)
        o1 = tr.step_posenet(f["img"][s], f["cloud"][s], f["choose"][s], f["obj"][s], f["target"][s], f["model_points"][s], sym[s], 0.015, dropout=False)
        _close(o1["loss"], a["loss"][s], 1e-5, "loss"); _close(o1["new_points"], a["new_points"][s], 1e-5, "new_points")
    _close(tr.grad, g_batched, 5e-4, "flat gradient")
    runs = []
    for _ in range(2):
        tr.zero_grad()
        tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=True, seed=77)
        runs.append(tr.grad.clone())
    assert torch.equal(runs[0], runs[1]) and float(runs[0].abs().sum()) > 0
    tr.zero_grad()
    tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=True, seed=78)
    assert not torch.equal(tr.grad, runs[0])                      # another seed, other masks
    # refiner: same two statements
    trr = _trainer("refiner", N, K, synth.make_state_dict(synth.refiner_spec(K), 1017))
    emb = a["emb"]
    r1 = trr.step_refiner(a["new_points"], emb, f["obj"], a["new_target"], f["model_points"], sym)
    gb = trr.grad.clone()
    trr.zero_grad()
    for b in range(B):
        s = slice(b, b + 1)
        trr.step_refiner(a["new_points"][s], emb[s], f["obj"][s], a["new_target"][s], f["model_points"][s], sym[s])
    _close(trr.grad, gb, 5e-4, "refiner flat gradient")
    trr.zero_grad()
    trr.step_refiner(a["new_points"], emb, f["obj"], a["new_target"], f["model_points"], sym)
    assert torch.equal(trr.grad, gb) and float(r1["dis"].min()) > 0


def test_native_step_at_the_ycb_training_shape_with_adam_and_a_hipgraph():
    """BASELINE configs[3] sizes (K = 21, N = 1000, M = 500, a symmetric object: the 250 M-pair nearest-neighbour branch): the
    native step agrees with the autograd-tape training path (densefusion_amd/lib/train_graph.py, itself held to the oracle and
    to the reference's backward), an Adam update on the flat buffers changes the next loss, and the whole step replays from a
    hipGraph with the gradients it produced eagerly."""
    from densefusion_amd import train_utils
    from densefusion_amd.lib import train_graph
    from densefusion_amd.lib.loss import Loss
    from densefusion_amd.lib.network import PoseNet
    K, N, H, W, M = 21, 1000, 80, 120, 500
    sym_list = [12, 15, 18, 19, 20]
    sd = synth.make_state_dict(synth.posenet_spec(K), 13)
    objs = [synth.make_object(105 + i, H, W, N, K, num_points_mesh=M) for i in range(2)]
    objs[0]["obj"][0], objs[1]["obj"][0] = 15, 3
    sym = [True, False]
    f = _frames(objs)
    net = PoseNet(N, K)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(DEV).train()
    r, t, c, _ = train_graph.posenet_forward(net, f["img"], f["cloud"], f["choose"], f["obj"], dropout=False)
    crit = Loss(M, sym_list)
    total = 0
    for b in range(2):
        ob = train_utils.with_host_index(f["obj"][b:b + 1], objs[b]["obj"])
        total = total + crit(r[b:b + 1], t[b:b + 1], c[b:b + 1], f["target"][b:b + 1], f["model_points"][b:b + 1], ob, f["cloud"][b:b + 1], 0.015, False)[0]
    total.backward()
    tr = _trainer("posenet", N, K, sd)
    out = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=False)
    _close(out["loss"].sum(), total, 1e-4, "loss")
    got = tr.grad_dict()
    for key, p in net.named_parameters():
        if "classifier" in key:
            continue
        _close(got[key], p.grad, 2e-3, key)
    eager = tr.grad.clone()
    # the same step from a hipGraph (graph_safe: the data-gradient weight copies are rebuilt inside the captured step)
    tr.zero_grad()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=False, graph_safe=True)
    torch.cuda.current_stream().wait_stream(side)
    tr.zero_grad()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        captured = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=False,
                                   graph_safe=True)
    tr.zero_grad()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(tr.grad, eager)
    _close(captured["loss"], out["loss"], 1e-6, "replayed loss")
    # a captured step would replay ONE set of Dropout2d masks for ever (the seed is a by-value launch argument): refused unless asked for
    with pytest.raises(RuntimeError, match="freeze the Dropout2d masks"):
        tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=True, graph_safe=True)
    # default seeds: a hash of (torch seed, lane, call) -- no two equal across lanes and calls
    a, b = _trainer("posenet", N, K, sd), _trainer("posenet", N, K, sd)
    b._salt = 1
    seeds = [t._next_seed() for _ in range(2000) for t in (a, b)]
    assert len(set(seeds)) == len(seeds)
    # one Adam step on the flat buffers (kernel layout: Adam is element-wise) lowers this batch's loss
    opt = train_utils.FlatAdam(tr, lr=1e-4)
    v0 = tr.version
    opt.step(grad_scale=0.5)
    assert tr.version == v0 + 1
    tr.zero_grad()
    out2 = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=False)
    assert float(out2["loss"].sum()) < float(out["loss"].sum())


def test_lanes_give_the_single_lane_gradients_and_repeat_bit_for_bit():
    """native_train.Lanes: the bs = 1 passes of a window on 3 concurrent lanes (streams + host threads + own gradient buffers)
    == the same passes one after the other, up to the summation order; a fixed lane count repeats bit for bit."""
    from densefusion_amd.native_train import Lanes
    K, N, H, W, M, B = 3, 128, 40, 80, 60, 5
    sd = synth.make_state_dict(synth.posenet_spec(K), 19)
    objs = [synth.make_object(400 + i, H if i % 2 else 80, W, N, K, num_points_mesh=M) for i in range(B)]      # two crop sizes in one window
    for i, o in enumerate(objs):
        o["obj"][0] = i % K
    fr = [_frames([o]) for o in objs]
    sym = [int(o["obj"][0]) == 1 for o in objs]
    tr = _trainer("posenet", N, K, sd)
    job = lambda i: (lambda lane: lane.step_posenet(fr[i]["img"], fr[i]["cloud"], fr[i]["choose"], fr[i]["obj"], fr[i]["target"], fr[i]["model_points"],
                                                   [sym[i]], 0.015, dropout=False)["dis"])
    single = [job(i)(tr) for i in range(B)]
    g1 = tr.grad.clone()
    lanes = Lanes(tr, 3)
    runs = []
    for _ in range(2):
        tr.zero_grad()
        dis = lanes.run([job(i) for i in range(B)])
        torch.cuda.synchronize()
        runs.append(tr.grad.clone())
        for a, b in zip(dis, single):
            assert torch.equal(a, b)
    lanes.close()
    assert torch.equal(runs[0], runs[1])
    _close(runs[0], g1, 5e-4, "lane-summed gradient")


def test_native_step_with_repeated_pixels_and_a_repeated_object():
    """Corners of the backward pass: `choose` wrap-padded from a few mask pixels (datasets/ycb/dataset.py:156-163: every pixel is
    chosen several times, so the chosen-pixel adjoints add several points into one pixel) and two frames of the SAME object in one
    pass (their last-layer gradients land in the same rows) -- against torch CPU autograd through the oracle, frame by frame."""
    K, N, H, W, M = 3, 64, 40, 40, 60
    sd = synth.make_state_dict(synth.posenet_spec(K), 23)
    objs = [synth.make_object(800 + i, H, W, N, K, num_points_mesh=M) for i in range(2)]
    rng = np.random.default_rng(5)
    for o in objs:
        o["obj"][0] = 2
        few = np.sort(rng.choice(H * W, size=11, replace=False))
        o["choose"] = np.resize(few, N).reshape(1, N).astype(np.int64)              # np.pad(..., 'wrap') of 11 pixels
    psd = {k: torch.from_numpy(v).clone().requires_grad_() for k, v in sd.items()}
    total = 0
    for o in objs:
        T = lambda k: torch.from_numpy(o[k])[None]
        idx = torch.tensor([[2]])
        r, t, c, _ = dfnet.posenet_forward(psd, T("img"), T("cloud"), torch.from_numpy(o["choose"]), idx)
        total = total + loss_ref.loss_calculation(r, t, c, T("target"), T("model_points"), idx, T("cloud"), 0.015, False, M, [2])[0]
    total.backward()
    tr = _trainer("posenet", N, K, sd)
    f = _frames(objs)
    out = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [True, True], 0.015, dropout=False)
    _close(out["loss"].sum(), total, 1e-4, "loss")
    for key, g in tr.grad_dict().items():
        if "classifier" in key:
            continue
        _close(g, psd[key].grad, 3e-3, key)


def test_native_steps_at_the_linemod_training_shape():
    """LineMOD sizes (K = 13, N = 500 -- padded to 512 point rows inside the step --, M = 500, the symmetric 'eggbox' index 7): the
    PoseNet step and the refiner step agree with the autograd-tape path at that shape too (the tape path is held to the oracle and
    to the reference's backward)."""
    from densefusion_amd import train_utils
    from densefusion_amd.lib import train_graph
    from densefusion_amd.lib.loss import Loss
    from densefusion_amd.lib.loss_refiner import Loss_refine
    from densefusion_amd.lib.network import PoseNet, PoseRefineNet
    K, N, H, W, M = 13, 500, 120, 80, 500
    sym_list = [7, 8]
    sd = synth.make_state_dict(synth.posenet_spec(K), 31)
    objs = [synth.make_object(305 + i, H, W, N, K, num_points_mesh=M) for i in range(2)]
    objs[0]["obj"][0], objs[1]["obj"][0] = 7, 2
    sym = [True, False]
    f = _frames(objs)
    net = PoseNet(N, K)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(DEV).train()
    r, t, c, emb = train_graph.posenet_forward(net, f["img"], f["cloud"], f["choose"], f["obj"], dropout=False)
    crit = Loss(M, sym_list)
    total, new_pts, new_tgt = 0, [], []
    for b in range(2):
        ob = train_utils.with_host_index(f["obj"][b:b + 1], objs[b]["obj"])
        loss, _, npt, ntg = crit(r[b:b + 1], t[b:b + 1], c[b:b + 1], f["target"][b:b + 1], f["model_points"][b:b + 1], ob, f["cloud"][b:b + 1], 0.015, False)
        total = total + loss
        new_pts.append(npt); new_tgt.append(ntg)
    total.backward()
    tr = _trainer("posenet", N, K, sd)
    out = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], sym, 0.015, dropout=False)
    _close(out["loss"].sum(), total, 1e-4, "loss")
    _close(out["new_points"], torch.cat(new_pts), 1e-4, "new_points")
    got = tr.grad_dict()
    for key, p in net.named_parameters():
        if "classifier" not in key:
            _close(got[key], p.grad, 2e-3, key)
    # refiner step on the re-centred points (tools/train.py:156-159)
    rsd = synth.make_state_dict(synth.refiner_spec(K), 32)
    ref = PoseRefineNet(N, K)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in rsd.items()})
    ref.to(DEV).train()
    pts, tgt, e = torch.cat(new_pts).detach(), torch.cat(new_tgt).detach(), emb.detach()
    pr, pt = train_graph.refiner_forward(ref, pts, e, f["obj"])
    crit_r = Loss_refine(M, sym_list)
    total_r = 0
    for b in range(2):
        ob = train_utils.with_host_index(f["obj"][b:b + 1], objs[b]["obj"])
        total_r = total_r + crit_r(pr[b:b + 1], pt[b:b + 1], tgt[b:b + 1], f["model_points"][b:b + 1], ob, pts[b:b + 1])[0]
    total_r = total_r.reshape(())
    total_r.backward()
    rt = _trainer("refiner", N, K, rsd)
    ro = rt.step_refiner(pts, e, f["obj"], tgt, f["model_points"], sym)
    _close(ro["dis"].sum(), total_r, 1e-4, "dis")
    rgot = rt.grad_dict()
    for key, p in ref.named_parameters():
        _close(rgot[key], p.grad, 2e-3, key)


def test_refiner_step_at_the_ycb_refine_mesh_size():
    """The refiner phase of YCB training samples 2600 model points (datasets/ycb/dataset.py:90-91,240-244): the symmetric branch is
    then a 2600 x 2600 nearest-neighbour search per frame.  Native refiner step vs torch CPU autograd through the oracle, two frames
    (one symmetric, one not)."""
    K, N, M = 21, 1000, 2600
    sd = synth.make_state_dict(synth.refiner_spec(K), 1013)
    objs = [synth.make_object(403 + i, 80, 80, N, K, num_points_mesh=M) for i in range(2)]
    objs[0]["obj"][0], objs[1]["obj"][0] = 19, 4
    rng = np.random.default_rng(2)
    emb = torch.from_numpy(rng.standard_normal((2, 32, N)).astype(np.float32))
    psd = {k: torch.from_numpy(v).clone().requires_grad_() for k, v in sd.items()}
    want, want_np, want_nt = 0, [], []
    for b, o in enumerate(objs):
        T = lambda k: torch.from_numpy(o[k])[None]
        idx = torch.tensor([[int(o["obj"][0])]])
        pr, pt = dfnet.refiner_forward(psd, T("cloud"), emb[b:b + 1], idx)
        d, npt, ntg = loss_ref.loss_refine_calculation(pr, pt, T("target"), T("model_points"), idx, T("cloud"), M, [12, 15, 18, 19, 20])
        want = want + d.reshape(())
        want_np.append(npt); want_nt.append(ntg)
    want.backward()
    f = _frames(objs)
    tr = _trainer("refiner", N, K, sd)
    out = tr.step_refiner(f["cloud"], emb.to(DEV), f["obj"], f["target"], f["model_points"], [True, False])
    _close(out["dis"].sum(), want, 1e-4, "dis")
    _close(out["new_points"], torch.cat(want_np), 1e-4, "new_points"); _close(out["new_target"], torch.cat(want_nt), 1e-4, "new_target")
    for key, g in tr.grad_dict().items():
        _close(g, psd[key].grad, 2e-3, key)


def test_a_window_of_mixed_crop_sizes_as_one_pass_equals_the_sum_of_its_one_frame_passes():
    """df_posenet_train_step_multi: frames of DIFFERENT crop sizes in one pass (what real data gives, tools/train.py:131-176 of the
    reference trains on whatever crop each frame has).  NOT bit-identical to the frames' bs = 1 passes -- one contraction over all
    frames' pixels adds in another order than frame-by-frame accumulation -- but equal up to fp32 summation order:
      * with split-K off (df_trainer_set_splitk: every output element summed in one order whatever shares the pass) every parameter
        tensor's gradient is within 5e-6 of its largest entry of the sum of the one-frame passes (measured worst 4.3e-7; whole-buffer
        relative L2 1.1e-7), per-frame outputs (loss, dis, re-centred clouds, emb) within 2e-5;
      * with the default split-K the one-frame passes split their small grids and the window does not: the ReLU-gated backward
        amplifies that re-association to 4e-3 of a tensor's scale at worst (layer4.1.conv2: 900 pixels), 3e-4 relative L2 over the
        buffer -- the same level as B same-size frames per pass against one per pass (test above: 5e-4) -- bounded here at 2e-2 / 2e-3.
    Two identical multi-bucket passes are bit-identical, dropout on.  Covers a bucket with two frames, symmetric and plain objects, a crop
    the F(4x4,3x3) route takes (160 x 160) and ones it does not."""
    K, N, M = 3, 128, 60
    sizes = [(40, 80), (160, 160), (80, 80), (40, 80), (120, 160)]
    sd = synth.make_state_dict(synth.posenet_spec(K), 23)
    objs = [synth.make_object(900 + i, h, w, N, K, num_points_mesh=M) for i, (h, w) in enumerate(sizes)]
    for i, o in enumerate(objs):
        o["obj"][0] = i % K
    frames = [dict(img=torch.from_numpy(o["img"]).to(DEV), cloud=torch.from_numpy(o["cloud"]).to(DEV), choose=torch.from_numpy(o["choose"]).to(DEV),
                   obj=torch.from_numpy(o["obj"]).to(DEV), target=torch.from_numpy(o["target"]).to(DEV), model_points=torch.from_numpy(o["model_points"]).to(DEV),
                   symmetric=int(o["obj"][0]) == 1) for o in objs]
    tr = _trainer("posenet", N, K, sd)
    for splitk, tol_tensor, tol_l2 in ((False, 5e-6, 2e-6), (True, 2e-2, 2e-3)):
        tr.set_splitk(splitk)
        tr.zero_grad()
        out, order = tr.step_posenet_window(frames, 0.015, dropout=False)
        assert sorted(order) == list(range(len(frames))) and order[:2] == [0, 3]          # the two 40 x 80 frames share a bucket
        g_multi = tr.grad_dict()
        tr.zero_grad()
        for row, j in enumerate(order):
            f = frames[j]
            o1 = tr.step_posenet(f["img"][None], f["cloud"][None], f["choose"].reshape(1, -1), f["obj"].reshape(1), f["target"][None], f["model_points"][None],
                                 [f["symmetric"]], 0.015, dropout=False)
            for k in ("loss", "dis", "new_points", "new_target", "emb"):
                _close(out[k][row:row + 1], o1[k], 2e-5, k)
        g_single = tr.grad_dict()
        worst, num, den = 0.0, 0.0, 0.0
        for k, v in g_single.items():
            if "classifier" in k:
                assert float(g_multi[k].abs().max()) == 0.0          # dead weights (lib/pspnet.py:58-62) receive no gradient
                continue
            scale = max(float(v.abs().max()), 1e-12)
            err = float((g_multi[k] - v).abs().max()) / scale
            worst = max(worst, err)
            num += float(((g_multi[k] - v).double() ** 2).sum()); den += float((v.double() ** 2).sum())
            assert err <= tol_tensor, f"split-K {splitk}: {k}: window vs sum of one-frame passes {err:.2e} of the tensor's scale"
        l2 = (num / den) ** 0.5
        print(f"mixed window vs one-frame passes, split-K {splitk}: worst per-tensor deviation {worst:.2e} of scale, relative L2 {l2:.2e}")
        assert l2 <= tol_l2
    runs = []
    for _ in range(2):
        tr.zero_grad()
        tr.step_posenet_window(frames, 0.015, dropout=True, seed=5)
        runs.append(tr.grad.clone())
    assert torch.equal(runs[0], runs[1]) and float(runs[0].abs().sum()) > 0


def test_trainer_profile_reports_executed_flops_per_kind():
    K, N, M = 3, 128, 60
    sd = synth.make_state_dict(synth.posenet_spec(K), 29)
    o = synth.make_object(950, 80, 80, N, K, num_points_mesh=M)
    f = _frames([o])
    tr = _trainer("posenet", N, K, sd)
    tr.profile(True)
    tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [False], 0.015, dropout=False)
    torch.cuda.synchronize()
    prof = tr.profile_read()
    tr.profile(False)
    assert set(prof) == {"fwd", "dgrad", "wgrad"}
    for kind, (ms, fl, n) in prof.items():
        assert ms > 0 and fl > 0 and n > 10, (kind, ms, fl, n)
    # the weight gradients contract exactly the forward launches' shapes except the Winograd-domain ones (fewer multiplies forward)
    assert prof["wgrad"][1] >= prof["fwd"][1] * 0.9


def test_crops_that_are_not_multiples_of_8_through_the_training_step():
    """The datasets snap crops to multiples of 40, the entry points accept any H, W >= 8: a 44 x 52 frame against the oracle's autograd (forward
    outputs, loss and every parameter gradient), and a window of 44 x 52 / 92 x 108 / 40 x 80 frames as one multi-bucket pass against its one-frame
    passes (split-K off: one summation order per output element)."""
    K, N, M = 2, 64, 60
    sd = synth.make_state_dict(synth.posenet_spec(K), 29)
    o = synth.make_object(1201, 44, 52, N, K, num_points_mesh=M)
    o["obj"][0] = 1
    idx = torch.tensor([[1]])
    T = lambda k: torch.from_numpy(o[k])[None]
    psd = {k: torch.from_numpy(v).clone().requires_grad_() for k, v in sd.items()}
    r, t, c, emb = dfnet.posenet_forward(psd, T("img"), T("cloud"), torch.from_numpy(o["choose"]), idx)
    want_loss, want_dis, _, _ = loss_ref.loss_calculation(r, t, c, T("target"), T("model_points"), idx, T("cloud"), 0.015, False, M, [1])
    want_loss.backward()
    tr = _trainer("posenet", N, K, sd)
    f = _frames([o])
    out = tr.step_posenet(f["img"], f["cloud"], f["choose"], f["obj"], f["target"], f["model_points"], [True], 0.015, dropout=False, want_pred=True)
    _close(out["pred_r"], r, 2e-4, "pred_r"); _close(out["emb"], emb, 2e-4, "emb"); _close(out["loss"], want_loss.reshape(1), 1e-4, "loss")
    for key, g in tr.grad_dict().items():
        if "classifier" not in key:
            _close(g, psd[key].grad, 2e-3, key)
    sizes = [(44, 52), (92, 108), (40, 80), (44, 52)]
    objs = [synth.make_object(1300 + i, h, w, N, K, num_points_mesh=M) for i, (h, w) in enumerate(sizes)]
    frames = [dict(img=torch.from_numpy(q["img"]).to(DEV), cloud=torch.from_numpy(q["cloud"]).to(DEV), choose=torch.from_numpy(q["choose"]).to(DEV),
                   obj=torch.from_numpy(q["obj"]).to(DEV), target=torch.from_numpy(q["target"]).to(DEV), model_points=torch.from_numpy(q["model_points"]).to(DEV),
                   symmetric=int(q["obj"][0]) == 1) for q in objs]
    tr.set_splitk(False)
    tr.zero_grad()
    outw, order = tr.step_posenet_window(frames, 0.015, dropout=False)
    g_multi = tr.grad_dict()
    tr.zero_grad()
    for row, j in enumerate(order):
        q = frames[j]
        o1 = tr.step_posenet(q["img"][None], q["cloud"][None], q["choose"].reshape(1, -1), q["obj"].reshape(1), q["target"][None], q["model_points"][None],
                             [q["symmetric"]], 0.015, dropout=False)
        for k in ("loss", "dis", "emb"):
            _close(outw[k][row:row + 1], o1[k], 2e-5, k)
    for k, v in tr.grad_dict().items():
        if "classifier" in k:
            continue
        scale = max(float(v.abs().max()), 1e-12)
        assert float((g_multi[k] - v).abs().max()) / scale <= 5e-6, k
