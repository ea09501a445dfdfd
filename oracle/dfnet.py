"""Oracle: functional CPU restatement of PoseNet / PoseRefineNet forward (torch, fp32 or fp64).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Works on a flat ``{key: tensor}`` state dict
with the reference's checkpoint keys; no nn.Module tree, no autograd.

Follows: lib/extractors.py:29-43,99-124 (ResNet-18, no BN, dilated layer3/4),
lib/pspnet.py:20-24,27-37,64-77 (PSP pyramid, x2 upsample convs, 1x1 + LogSoftmax),
lib/network.py:53-68,95-132 (PoseNet), :151-168,187-206 (PoseRefineNet).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

_CNN = "cnn.model.module."


def _to_torch_sd(sd, dtype=torch.float32):
    return {k: torch.as_tensor(v).to(dtype) for k, v in sd.items()}


def _basic_block(sd, base, x, stride, dilation):
    # lib/extractors.py:29-43 -- conv3x3 -> relu -> conv3x3 -> (+residual|downsample) -> relu
    out = F.conv2d(x, sd[base + "conv1.weight"], None, stride=stride, padding=dilation, dilation=dilation)
    out = F.relu(out)
    out = F.conv2d(out, sd[base + "conv2.weight"], None, stride=1, padding=dilation, dilation=dilation)
    key = base + "downsample.0.weight"
    res = F.conv2d(x, sd[key], None, stride=stride) if key in sd else x
    return F.relu(out + res)


def resnet18_forward(sd, img, taps=None):
    # lib/extractors.py:114-124; layer strides/dilations from :88-91; the first block of every
    # layer is built WITHOUT dilation (:107), later blocks with it (:110)
    p = _CNN + "feats."
    x = F.relu(F.conv2d(img, sd[p + "conv1.weight"], None, stride=2, padding=3))
    if taps is not None:
        taps["stem"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (stride, dil) in enumerate(((1, 1), (2, 1), (1, 2), (1, 4)), start=1):
        x = _basic_block(sd, f"{p}layer{li}.0.", x, stride, 1)
        x = _basic_block(sd, f"{p}layer{li}.1.", x, 1, dil)
        if taps is not None:
            taps[f"layer{li}"] = x
    return x


def psp_forward(sd, feats, taps=None):
    # lib/pspnet.py:20-24 -- AdaptiveAvgPool(s) -> 1x1 conv (no bias) -> bilinear to (h,w) with
    # the F.upsample default align_corners=False; cat priors + feats; 1x1 bottleneck; relu
    p = _CNN + "psp."
    h, w = feats.shape[2], feats.shape[3]
    priors = []
    for i, s in enumerate((1, 2, 3, 6)):
        y = F.adaptive_avg_pool2d(feats, (s, s))
        y = F.conv2d(y, sd[f"{p}stages.{i}.1.weight"], None)
        priors.append(F.interpolate(y, size=(h, w), mode="bilinear", align_corners=False))
    priors.append(feats)
    out = F.conv2d(torch.cat(priors, 1), sd[p + "bottleneck.weight"], sd[p + "bottleneck.bias"])
    return F.relu(out)


def _psp_upsample(sd, name, x):
    # lib/pspnet.py:27-37 -- nn.Upsample(x2, bilinear, align_corners=True) -> conv3x3 p1 -> PReLU
    p = f"{_CNN}{name}.conv."
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.conv2d(x, sd[p + "1.weight"], sd[p + "1.bias"], padding=1)
    return F.prelu(x, sd[p + "2.weight"])


def pspnet_forward(sd, img, taps=None):
    # lib/pspnet.py:64-77 (dropout layers are identity in eval mode)
    f = resnet18_forward(sd, img, taps)
    p = psp_forward(sd, f, taps)
    if taps is not None:
        taps["psp"] = p
    for name in ("up_1", "up_2", "up_3"):
        p = _psp_upsample(sd, name, p)
        if taps is not None:
            taps[name] = p
    p = F.conv2d(p, sd[_CNN + "final.0.weight"], sd[_CNN + "final.0.bias"])
    return F.log_softmax(p, dim=1)     # nn.LogSoftmax() implicit dim -> 1 for 4-D input


def _pc(sd, key, x, relu=True):
    y = F.conv1d(x, sd[key + ".weight"], sd[key + ".bias"])
    return F.relu(y) if relu else y


def posenet_feat(sd, x, emb, taps=None):
    # lib/network.py:53-68
    x1 = _pc(sd, "feat.conv1", x)
    e1 = _pc(sd, "feat.e_conv1", emb)
    x2 = _pc(sd, "feat.conv2", x1)
    e2 = _pc(sd, "feat.e_conv2", e1)
    pf1 = torch.cat((x1, e1), 1)
    pf2 = torch.cat((x2, e2), 1)
    x5 = _pc(sd, "feat.conv5", pf2)
    x6 = _pc(sd, "feat.conv6", x5)
    n = x.shape[2]
    ap = F.avg_pool1d(x6, n)                       # AvgPool1d(num_points)
    if taps is not None:
        taps["ap_x"] = ap.reshape(-1, 1024)
    return torch.cat([pf1, pf2, ap.reshape(-1, 1024, 1).repeat(1, 1, n)], 1)


def posenet_forward(sd, img, x, choose, obj, taps=None):
    """PoseNet.forward for ONE object (bs = 1 semantics, lib/network.py:95-132).

    img [1,3,H,W]; x [1,N,3]; choose [1,1,N] or [1,N] int64; obj [1,1] or [1] int64.
    Returns out_rx [1,N,4], out_tx [1,N,3], out_cx [1,N,1], emb [1,32,N].
    """
    out_img = pspnet_forward(sd, img, taps)
    bs, di = out_img.shape[0], out_img.shape[1]
    n = x.shape[1]
    emb = out_img.reshape(bs, di, -1)
    idx = choose.reshape(bs, 1, n).repeat(1, di, 1)
    emb = torch.gather(emb, 2, idx).contiguous()
    xt = x.transpose(2, 1).contiguous()
    ap_x = posenet_feat(sd, xt, emb, taps)
    outs = {}
    for h in "rtc":
        y = _pc(sd, f"conv1_{h}", ap_x)
        y = _pc(sd, f"conv2_{h}", y)
        y = _pc(sd, f"conv3_{h}", y)
        outs[h] = _pc(sd, f"conv4_{h}", y, relu=False)
    k = outs["c"].shape[1]
    o = int(obj.reshape(-1)[0])
    rx = outs["r"].reshape(bs, k, 4, n)[0, o]            # [4,N]
    tx = outs["t"].reshape(bs, k, 3, n)[0, o]
    cx = torch.sigmoid(outs["c"]).reshape(bs, k, 1, n)[0, o]
    return (rx.t().contiguous()[None], tx.t().contiguous()[None], cx.t().contiguous()[None], emb)


def refiner_feat(sd, x, emb):
    # lib/network.py:151-168
    x1 = _pc(sd, "feat.conv1", x)
    e1 = _pc(sd, "feat.e_conv1", emb)
    x2 = _pc(sd, "feat.conv2", x1)
    e2 = _pc(sd, "feat.e_conv2", e1)
    pf3 = torch.cat([x1, e1, x2, e2], 1)
    x5 = _pc(sd, "feat.conv5", pf3)
    x6 = _pc(sd, "feat.conv6", x5)
    return F.avg_pool1d(x6, x.shape[2]).reshape(-1, 1024)


def refiner_forward(sd, x, emb, obj):
    """PoseRefineNet.forward for ONE object (lib/network.py:187-206): x [1,N,3], emb [1,32,N]."""
    ap = refiner_feat(sd, x.transpose(2, 1).contiguous(), emb)
    outs = {}
    for h in "rt":
        y = F.relu(F.linear(ap, sd[f"conv1_{h}.weight"], sd[f"conv1_{h}.bias"]))
        y = F.relu(F.linear(y, sd[f"conv2_{h}.weight"], sd[f"conv2_{h}.bias"]))
        outs[h] = F.linear(y, sd[f"conv3_{h}.weight"], sd[f"conv3_{h}.bias"])
    o = int(obj.reshape(-1)[0])
    k = outs["t"].shape[1] // 3
    return outs["r"].reshape(1, k, 4)[0, o][None], outs["t"].reshape(1, k, 3)[0, o][None]
