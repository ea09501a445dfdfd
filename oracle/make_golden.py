"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules on CPU.

Dev-only script, run in the build container where /root/reference exists; the GPU box only sees
the committed .npz fixtures (inputs are regenerated there from seeds by densefusion_amd.synth).
Nothing from the reference is copied: the modules are imported from where they lie, given the
build's synthetic weights through load_state_dict(strict=True), and their outputs are stored.

The reference imports torchvision (unused on this path, lib/network.py:10-11), which is not
installed here; empty placeholder modules are registered so the import proceeds.

    python -m oracle.make_golden            # writes tests/golden/
"""
from __future__ import annotations

import copy
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densefusion_amd import synth  # noqa: E402

REF = os.environ.get("DF_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

# (name, K, N, H, W, refine iterations, weight seed, input seed, cam)
CASES = [
    ("tiny", 2, 64, 40, 40, 2, 11, 101, "ycb"),
    ("cfg1_linemod_80", 13, 500, 80, 80, 0, 12, 102, "linemod"),
    ("cfg2_linemod_120x160", 13, 500, 120, 160, 2, 12, 103, "linemod"),
    ("cfg3_ycb_160", 21, 1000, 160, 160, 4, 13, 104, "ycb"),
    ("cfg3_ycb_80x120", 21, 1000, 80, 120, 2, 13, 105, "ycb"),
    # large / stress shapes (outputs only): the biggest bench crop, a full frame, and BASELINE configs[4]'s 2000 points
    ("cfg3_ycb_240x320", 21, 1000, 240, 320, 2, 13, 106, "ycb"),
    ("cfg3_ycb_480x640", 21, 1000, 480, 640, 2, 13, 107, "ycb"),
    ("cfg5_n2000_240x320", 21, 2000, 240, 320, 2, 13, 108, "ycb"),
]


def import_reference():
    for m in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    import lib.network as network
    import lib.loss as loss
    import lib.loss_refiner as loss_refiner
    import lib.nn as nnmod
    import lib.transformations as tf
    return network, loss, loss_refiner, nnmod, tf


def t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def run_case(ref, name, K, N, H, W, iters, wseed, iseed, cam):
    network, _, _, _, tf = ref
    cam = synth.YCB_CAM if cam == "ycb" else synth.LINEMOD_CAM
    pspec, rspec = synth.posenet_spec(K), synth.refiner_spec(K)
    est = network.PoseNet(num_points=N, num_obj=K)
    rfn = network.PoseRefineNet(num_points=N, num_obj=K)
    assert [(k, tuple(v.shape)) for k, v in est.state_dict().items()] == pspec, "PoseNet key layout drifted"
    assert [(k, tuple(v.shape)) for k, v in rfn.state_dict().items()] == rspec, "PoseRefineNet key layout drifted"
    est.load_state_dict(t(synth.make_state_dict(pspec, wseed)), strict=True)
    rfn.load_state_dict(t(synth.make_state_dict(rspec, wseed + 1000)), strict=True)
    est.eval(); rfn.eval()
    # arg-max pose selection is discontinuous: pick an input seed whose top-2 confidence gap is
    # comfortably above fp32 noise, so a golden never sits on a tie
    while True:
        o = synth.make_object(iseed, H, W, N, K, cam=cam)
        img = torch.from_numpy(o["img"])[None]
        cloud = torch.from_numpy(o["cloud"])[None]
        choose = torch.from_numpy(o["choose"])
        index = torch.from_numpy(o["obj"])
        with torch.no_grad():
            cs = torch.sort(est(img, cloud, choose, index)[2].view(-1))[0]
        if float(cs[-1] - cs[-2]) > 5e-4:
            break
        iseed += 1000
    out = {}
    taps = {}
    if name == "tiny":
        m = est.cnn.model.module
        hooks = []
        for nm, mod in (("stem", m.feats.relu), ("layer1", m.feats.layer1), ("layer2", m.feats.layer2),
                        ("layer3", m.feats.layer3), ("layer4", m.feats.layer4), ("psp", m.psp),
                        ("up_1", m.up_1), ("up_2", m.up_2), ("up_3", m.up_3), ("ap_x", est.feat.ap1)):
            def mk(nm):
                def hook(_m, _i, o_):
                    taps.setdefault(nm, o_.detach().clone().numpy())   # stem relu fires first
                return hook
            hooks.append(mod.register_forward_hook(mk(nm)))
    with torch.no_grad():
        pred_r, pred_t, pred_c, emb = est(img, cloud, choose, index)
        out.update(out_rx=pred_r.numpy(), out_tx=pred_t.numpy(), out_cx=pred_c.numpy(), emb=emb.numpy())
        for k, v in taps.items():
            out["tap_" + k] = v
        # --- the eval loop, transcribed around the imported modules (tools/eval_ycb.py:193-229) ---
        pred_r = pred_r / torch.norm(pred_r, dim=2).view(1, N, 1)
        how_max, which_max = torch.max(pred_c.view(1, N), 1)
        pred_t = pred_t.view(N, 1, 3)
        points = cloud.view(N, 1, 3)
        my_r = pred_r[0][which_max[0]].view(-1).numpy()
        my_t = (points + pred_t)[which_max[0]].view(-1).numpy()
        out["which_max"] = np.array([int(which_max[0])])
        out["pose_wo_refine"] = np.append(my_r, my_t).astype(np.float64)
        poses = []
        for ite in range(iters):
            T = torch.from_numpy(my_t.astype(np.float32)).view(1, 3).repeat(N, 1).contiguous().view(1, N, 3)
            my_mat = tf.quaternion_matrix(my_r)
            R = torch.from_numpy(my_mat[:3, :3].astype(np.float32)).view(1, 3, 3)
            my_mat[0:3, 3] = my_t
            new_cloud = torch.bmm((cloud - T), R).contiguous()
            pr, pt = rfn(new_cloud, emb, index)
            if ite == 0:
                out["refine0_rx"], out["refine0_tx"] = pr.numpy(), pt.numpy()
            pr = pr.view(1, 1, -1)
            pr = pr / (torch.norm(pr, dim=2).view(1, 1, 1))
            my_mat_2 = tf.quaternion_matrix(pr.view(-1).numpy())
            my_mat_2[0:3, 3] = pt.view(-1).numpy()
            my_mat_final = np.dot(my_mat, my_mat_2)
            my_r_final = copy.deepcopy(my_mat_final)
            my_r_final[0:3, 3] = 0
            my_r = tf.quaternion_from_matrix(my_r_final, True)
            my_t = np.array([my_mat_final[0][3], my_mat_final[1][3], my_mat_final[2][3]])
            poses.append(np.append(my_r, my_t))
        if poses:
            out["poses_refined"] = np.stack(poses).astype(np.float64)
    out["meta"] = np.array([K, N, H, W, iters, wseed, iseed])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, "which_max", out["which_max"], "pose", out.get("poses_refined", out["pose_wo_refine"])[-1] if iters else out["pose_wo_refine"])


def run_loss(ref):
    """Loss / Loss_refine for a NON-symmetric idx (the symmetric branch raises in this fork) and
    nn_distance as the small-size pin for the 1-NN restatement."""
    _, loss, loss_refiner, nnmod, _ = ref
    rng = np.random.Generator(np.random.PCG64(77))
    N, M = 96, 80
    q = rng.standard_normal((1, N, 4)).astype(np.float32)
    pt = (rng.standard_normal((1, N, 3)) * 0.05).astype(np.float32)
    pc = rng.uniform(0.05, 0.95, (1, N, 1)).astype(np.float32)
    o = synth.make_object(555, 80, 80, N, 13, num_points_mesh=M)
    target, mp, pts = o["target"][None], o["model_points"][None], o["cloud"][None]
    idx = torch.tensor([[3]])
    crit = loss.Loss(M, [7, 8])
    l, d, npnts, ntgt = crit(*(torch.from_numpy(a) for a in (q, pt, pc, target, mp)), idx,
                             torch.from_numpy(pts), 0.015, False)
    critr = loss_refiner.Loss_refine(M, [7, 8])
    q1 = rng.standard_normal((1, 4)).astype(np.float32)
    t1 = (rng.standard_normal((1, 3)) * 0.02).astype(np.float32)
    d2, np2, nt2 = critr(torch.from_numpy(q1), torch.from_numpy(t1), ntgt, torch.from_numpy(mp), idx, npnts)
    # 1-NN pin: knn(ref=target[1,3,R], query=pred[1,3,Q]) == nn_distance(pred^T, target^T)[1] + 1
    R_, Q_ = 70, 333
    refp = rng.random((2, 3, R_)).astype(np.float32)
    qry = rng.random((2, 3, Q_)).astype(np.float32)
    _, idx1, _, _ = nnmod.nn_distance(torch.from_numpy(qry).transpose(2, 1).contiguous(),
                                      torch.from_numpy(refp).transpose(2, 1).contiguous())
    np.savez_compressed(os.path.join(OUT, "loss_nonsym.npz"),
                        pred_r=q, pred_t=pt, pred_c=pc, target=target, model_points=mp, points=pts,
                        loss=l.numpy(), dis=d.numpy(), new_points=npnts.numpy(), new_target=ntgt.numpy(),
                        r_pred_r=q1, r_pred_t=t1, r_dis=d2.numpy(), r_new_points=np2.numpy(),
                        r_new_target=nt2.numpy())
    np.savez_compressed(os.path.join(OUT, "nn_distance_small.npz"), ref=refp, query=qry,
                        idx_1based=(idx1.numpy() + 1).astype(np.int64))
    print("loss", float(l), float(d), float(d2))


def grad_sample(g, cap=8192):
    """the whole tensor when small, else every (numel // cap)-th element of the flattened tensor"""
    g = np.asarray(g)
    return g if g.size <= cap else g.reshape(-1)[::g.size // cap].copy()


def run_grad(ref):
    """Gradient golden: the imported reference's Loss (non-symmetric idx; the symmetric branch raises in this fork)
    .backward() through the imported PoseNet at the tiny config (tools/train.py:152-161), with the modules in eval()
    mode so that Dropout2d (lib/pspnet.py:46,52) is the identity and the result is deterministic.  Stored: loss, dis and a
    representative set of parameter gradients (first / middle / last layers of every part of the graph)."""
    network, loss, _, _, _ = ref
    K, N, H, W, M, wseed, iseed = 2, 64, 40, 40, 60, 11, 101
    est = network.PoseNet(num_points=N, num_obj=K)
    est.load_state_dict(t(synth.make_state_dict(synth.posenet_spec(K), wseed)), strict=True)
    est.eval()
    o = synth.make_object(iseed, H, W, N, K, num_points_mesh=M)
    idx = torch.tensor([[0]])
    T = lambda k: torch.from_numpy(o[k])[None]
    pred_r, pred_t, pred_c, emb = est(T("img"), T("cloud"), torch.from_numpy(o["choose"]), idx)
    crit = loss.Loss(M, [1])
    l, d, _, _ = crit(pred_r, pred_t, pred_c, T("target"), T("model_points"), idx, T("cloud"), 0.015, False)
    l.backward()
    keep = ["cnn.model.module.feats.conv1.weight", "cnn.model.module.feats.layer1.0.conv1.weight",
            "cnn.model.module.feats.layer2.0.downsample.0.weight", "cnn.model.module.feats.layer3.1.conv2.weight",
            "cnn.model.module.feats.layer4.1.conv1.weight", "cnn.model.module.psp.stages.2.1.weight",
            "cnn.model.module.psp.bottleneck.weight", "cnn.model.module.psp.bottleneck.bias",
            "cnn.model.module.up_1.conv.1.weight", "cnn.model.module.up_2.conv.2.weight", "cnn.model.module.up_3.conv.1.bias",
            "cnn.model.module.final.0.weight", "feat.conv1.weight", "feat.e_conv2.weight", "feat.conv6.bias",
            "conv1_r.weight", "conv2_t.weight", "conv3_c.bias", "conv4_r.weight", "conv4_c.weight"]
    grads = dict(est.named_parameters())
    # big tensors are stored as a fixed strided sample of their flattened gradient (grad_sample below; the test takes the same)
    out = {"grad:" + k: grad_sample(grads[k].grad.numpy()) for k in keep}
    out.update(loss=l.detach().numpy(), dis=d.detach().numpy(), out_rx=pred_r.detach().numpy(), out_cx=pred_c.detach().numpy(),
               meta=np.array([K, N, H, W, M, wseed, iseed, 0]))
    np.savez_compressed(os.path.join(OUT, "grad_tiny.npz"), **out)
    print("grad", float(l), float(d), {k[5:]: float(np.abs(v).max()) for k, v in out.items() if k.startswith("grad:")})


def run_quat(ref):
    tf = ref[4]
    rng = np.random.Generator(np.random.PCG64(5))
    qs = rng.standard_normal((64, 4))
    qs[:8] *= 1e-3
    mats = np.stack([tf.quaternion_matrix(q) for q in qs])
    back = np.stack([tf.quaternion_from_matrix(m, True) for m in mats])
    # reference doctest values, lib/transformations.py:1257-1258, 1302-1303
    doc_q = np.array([0.99810947, 0.06146124, 0, 0])
    doc_M = tf.quaternion_matrix(doc_q)
    Rdoc = tf.rotation_matrix(0.123, (1, 2, 3))
    np.savez_compressed(os.path.join(OUT, "quaternion.npz"), q=qs, M=mats, q_back=back,
                        doc_q=doc_q, doc_M=doc_M, doc_R123=Rdoc,
                        doc_q123=np.array([0.9981095, 0.0164262, 0.0328524, 0.0492786]))
    print("quat ok")


def read_ply_vertices(path):
    """binary_little_endian PLY with `element vertex N` of 3 doubles (written by tools/eval_cad.py:130-136)."""
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    assert b"property double x" in head
    return np.frombuffer(body[:n * 24], dtype="<f8").reshape(n, 3).copy()


def run_ply():
    pred = read_ply_vertices(os.path.join(REF, "pred_pcld_output.ply"))
    tgt = read_ply_vertices(os.path.join(REF, "target_pcld_output.ply"))
    np.savez_compressed(os.path.join(OUT, "ply_clouds.npz"), pred=pred, target=tgt)
    print("ply", pred.shape, tgt.shape)


def run_segnet():
    """vanilla_segmentation/segnet.py imported as it stands (pure torch), seeded synthetic weights incl. BatchNorm running
    statistics (synth.make_segnet_state_dict, loaded strict=True: proves the key layout), eval-mode logits of a 2 x 3 x 32 x 64
    normalised image batch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_segnet", os.path.join(REF, "vanilla_segmentation", "segnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = mod.SegNet()
    sd = synth.make_segnet_state_dict(77)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == synth.segnet_spec()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    net.eval()
    rng = np.random.Generator(np.random.PCG64(5))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[None, :, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[None, :, None, None]
    x = ((rng.integers(0, 256, (2, 3, 32, 64)).astype(np.float32) / 255.0 - mean) / std).astype(np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "segnet_small.npz"), x=x, logits=y.astype(np.float32), meta=np.array([77, 5]))
    print("segnet", y.shape)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref = import_reference()
    only = sys.argv[1:]
    for case in CASES:
        if not only or case[0] in only:
            run_case(ref, *case)
    if not only or "loss" in only:
        run_loss(ref)
    if not only or "grad" in only:
        run_grad(ref)
    if not only or "quat" in only:
        run_quat(ref)
    if not only or "ply" in only:
        run_ply()
    if not only or "segnet" in only:
        run_segnet()


if __name__ == "__main__":
    main()
