"""Differentiable (training-mode) forward of PoseNet / PoseRefineNet on the HIP kernels.

The inference engine (csrc/engine.hip) fuses and folds layers in ways that have no stored activations;
training instead walks the reference's layer graph (lib/extractors.py:114-124, lib/pspnet.py:64-77,
lib/network.py:53-68,95-132,151-206) once more.  torch.autograd is only the tape: every layer's forward and
backward is a HIP launch -- convolutions / Conv1d(k=1) / Linear with their bias, residual and ReLU / PReLU fused
on the fp32-MFMA kernels (``train_ops.ConvAct``: forward, activation gradient, data gradient, weight gradient),
pooling, bilinear resize, LogSoftmax, Dropout2d, the colour-feature gather, the mean over points and the sigmoid
through ``csrc/trainops.hip``.  What torch itself still does is tensor plumbing only: views, the channel
concatenations and zero-padding of the 3-channel inputs / 63-channel translation head.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import train_ops as T

_CNN = "cnn.model.module."


def _pad4(n):
    return (n + 3) // 4 * 4


def conv(x, w, b=None, stride=1, pad=0, dil=1, act=0, res=None, slope=None):
    """act(conv(x, w) + b + res).  x [B,H,W,Cin] channels-last; w in the REFERENCE layout [Cout,Cin,KH,KW] (or
    [Cout,Cin,1] / [Cout,Cin]); act 0 none, 1 ReLU, 2 PReLU(slope).  Channel counts that are not multiples of 4
    (3-channel image / cloud, 63 translation outputs) are zero-padded."""
    if w.dim() == 3:
        w = w.unsqueeze(-1)
    elif w.dim() == 2:
        w = w.unsqueeze(-1).unsqueeze(-1)
    cout, cin = w.shape[0], w.shape[1]
    w = w.permute(0, 2, 3, 1)                                   # -> [Cout,KH,KW,Cin]
    if cin % 4:
        w = F.pad(w, (0, _pad4(cin) - cin))
        x = F.pad(x, (0, _pad4(cin) - cin))
    if cout % 4:
        w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, _pad4(cout) - cout))
        if b is not None:
            b = F.pad(b, (0, _pad4(cout) - cout))
    y = T.ConvAct.apply(x, w.contiguous(), b.contiguous() if b is not None else None, res, slope, stride, pad, dil, act)
    return y[..., :cout] if cout % 4 else y


def _basic_block(P, base, x, stride, dilation):
    out = conv(x, P[base + "conv1.weight"], None, stride, dilation, dilation, act=1)
    key = base + "downsample.0.weight"
    res = conv(x, P[key], None, stride, 0, 1) if key in P else x
    return conv(out, P[base + "conv2.weight"], None, 1, dilation, dilation, act=1, res=res)     # relu(conv2 + residual)


def pspnet_forward(P, img, dropout, seed=0):
    """img [B,3,H,W] -> log-softmax colour features [B,H,W,32] (channels-last)."""
    f = _CNN + "feats."
    x = img.permute(0, 2, 3, 1).contiguous()
    x = conv(x, P[f + "conv1.weight"], None, 2, 3, 1, act=1)
    x = T.MaxPool3s2.apply(x)
    for li, (stride, dil) in enumerate(((1, 1), (2, 1), (1, 2), (1, 4)), start=1):
        x = _basic_block(P, f"{f}layer{li}.0.", x, stride, 1)
        x = _basic_block(P, f"{f}layer{li}.1.", x, 1, dil)
    h, w = x.shape[1], x.shape[2]
    p = _CNN + "psp."
    priors = []
    for i, s in enumerate((1, 2, 3, 6)):
        y = T.AdaptiveAvgPool.apply(x, s)
        y = conv(y, P[f"{p}stages.{i}.1.weight"])
        priors.append(T.Bilinear.apply(y, h, w, False))
    priors.append(x)
    x = conv(torch.cat(priors, dim=3), P[p + "bottleneck.weight"], P[p + "bottleneck.bias"], act=1)
    if dropout:
        x = T.Dropout2d.apply(x, 0.3, seed * 4 + 1)
    for k, name in enumerate(("up_1", "up_2", "up_3")):
        q = f"{_CNN}{name}.conv."
        x = T.Bilinear.apply(x, 2 * x.shape[1], 2 * x.shape[2], True)
        x = conv(x, P[q + "1.weight"], P[q + "1.bias"], 1, 1, 1, act=2, slope=P[q + "2.weight"])
        if dropout and name != "up_3":
            x = T.Dropout2d.apply(x, 0.15, seed * 4 + 2 + k)
    x = conv(x, P[_CNN + "final.0.weight"], P[_CNN + "final.0.bias"])
    return T.LogSoftmaxLast.apply(x)


def _pc(P, key, x, relu=True):
    """Conv1d(k=1) over points: x [1,N,1,C]."""
    return conv(x, P[key + ".weight"], P[key + ".bias"], act=1 if relu else 0)


_DROPOUT_CALLS = [0]


def _object_means(x6, B, N):
    """AvgPool1d(N) per object: x6 [B,N,1,C] -> [B,1,1,C]."""
    C = x6.shape[-1]
    return torch.stack([T.ColMean.apply(x6[b].reshape(N, C)) for b in range(B)]).reshape(B, 1, 1, C)


def posenet_forward(net, img, x, choose, obj, dropout=True):
    """Training-mode PoseNet.forward (lib/network.py:95-132), differentiable.  The reference runs one object per call
    (bs = 1); B same-size objects may be passed together here -- every layer is per-sample, so the outputs and the
    gradients are those of B separate calls."""
    P = dict(net.named_parameters())
    N = net.num_points
    B, _, H, W = img.shape
    _DROPOUT_CALLS[0] += 1
    feat = pspnet_forward(P, img, dropout, seed=int(torch.initial_seed() % 100003) * 7919 + _DROPOUT_CALLS[0])   # [B,H,W,32]
    idx = (choose.reshape(B, N) + torch.arange(B, device=choose.device).reshape(B, 1) * (H * W)).reshape(-1)
    emb_pm = T.GatherRows.apply(feat.reshape(-1, 32), idx)               # [B*N,32]  (gather at the chosen pixels)
    pts = x.reshape(B, N, 1, 3)
    e = emb_pm.reshape(B, N, 1, 32)
    x1, e1 = _pc(P, "feat.conv1", pts), _pc(P, "feat.e_conv1", e)
    x2, e2 = _pc(P, "feat.conv2", x1), _pc(P, "feat.e_conv2", e1)
    pf1, pf2 = torch.cat((x1, e1), 3), torch.cat((x2, e2), 3)
    x6 = _pc(P, "feat.conv6", _pc(P, "feat.conv5", pf2))
    ap = _object_means(x6, B, N).expand(B, N, 1, 1024)                    # AvgPool1d(N) + repeat
    ap_x = torch.cat((pf1, pf2, ap), 3)                                  # 128 + 256 + 1024
    outs = {}
    for hname in "rtc":
        y = _pc(P, f"conv1_{hname}", ap_x)
        y = _pc(P, f"conv2_{hname}", y)
        y = _pc(P, f"conv3_{hname}", y)
        outs[hname] = _pc(P, f"conv4_{hname}", y, relu=False).reshape(B, N, -1)
    # the object's slice of every head, picked on the device (indexing with the index TENSOR: no read-back, no synchronisation)
    bsel, osel = torch.arange(B, device=obj.device), obj.reshape(-1).long()
    out_rx = outs["r"].reshape(B, N, -1, 4)[bsel, :, osel]
    out_tx = outs["t"].reshape(B, N, -1, 3)[bsel, :, osel]
    conf = outs["c"].reshape(B, N, -1, 1)[bsel, :, osel]
    out_cx = T.Sigmoid.apply(conf.contiguous())
    emb = emb_pm.reshape(B, N, 32).transpose(1, 2).contiguous()
    return out_rx, out_tx, out_cx, emb.detach()


def refiner_forward(net, x, emb, obj):
    """Training-mode PoseRefineNet.forward (lib/network.py:187-206), differentiable; B objects per call allowed."""
    P = dict(net.named_parameters())
    N = net.num_points
    B = x.shape[0]
    pts = x.reshape(B, N, 1, 3)
    e = emb.reshape(B, 32, N).transpose(1, 2).reshape(B, N, 1, 32)
    x1, e1 = _pc(P, "feat.conv1", pts), _pc(P, "feat.e_conv1", e)
    x2, e2 = _pc(P, "feat.conv2", x1), _pc(P, "feat.e_conv2", e1)
    pf3 = torch.cat((x1, e1, x2, e2), 3)
    ap = _object_means(_pc(P, "feat.conv6", _pc(P, "feat.conv5", pf3)), B, N)
    outs = {}
    for hname in "rt":
        y = _pc(P, f"conv1_{hname}", ap)
        y = _pc(P, f"conv2_{hname}", y)
        outs[hname] = _pc(P, f"conv3_{hname}", y, relu=False).reshape(B, -1)
    bsel, osel = torch.arange(B, device=obj.device), obj.reshape(-1).long()
    out_rx = outs["r"].reshape(B, -1, 4)[bsel, osel]
    out_tx = outs["t"].reshape(B, -1, 3)[bsel, osel]
    return out_rx, out_tx
