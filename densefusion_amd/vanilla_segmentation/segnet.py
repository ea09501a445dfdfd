"""``SegNet`` -- host-side mirror of the reference's segmentation network (vanilla_segmentation/segnet.py:6-121), the
producer of the masks LineMOD evaluation reads (datasets/linemod/dataset.py:57-58).

Kept: ``SegNet(input_nbr=3, label_nbr=22)``, ``forward(x [B,3,H,W]) -> logits [B,label_nbr,H,W]``, and the state-dict
layout (``convXY.{weight,bias}``, ``bnXY.{weight,bias,running_mean,running_var,num_batches_tracked}`` for the 13
encoder and 13 decoder convolutions), so ``model.load_state_dict(torch.load(...))`` works on a reference checkpoint.

Different: the submodules are parameter containers only.  ``eval()`` forward folds every BatchNorm into its convolution
(w' = w g / sqrt(var + eps), b' = (b - mean) g / sqrt(var + eps) + beta), keeps activations channels-last and runs
conv + bias + ReLU as one launch of the fp32-MFMA kernel (the >= 256-channel layers through the Winograd domain), the
2x2 max-pool / un-pool pairs through ``df_maxpool2x2_idx`` / ``df_maxunpool2x2``.  H and W must be multiples of 32
(480 x 640 is).  Training this network is not rebuilt: ``train()`` mode forward raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops

# (name, Cin, Cout) in forward order; 'P' = pool, 'U' = unpool (segnet.py:73-118)
_ENC = [("11", None, 64), ("12", 64, 64), "P", ("21", 64, 128), ("22", 128, 128), "P", ("31", 128, 256), ("32", 256, 256), ("33", 256, 256), "P",
        ("41", 256, 512), ("42", 512, 512), ("43", 512, 512), "P", ("51", 512, 512), ("52", 512, 512), ("53", 512, 512), "P"]
_DEC = ["U", ("53d", 512, 512), ("52d", 512, 512), ("51d", 512, 512), "U", ("43d", 512, 512), ("42d", 512, 512), ("41d", 512, 256),
        "U", ("33d", 256, 256), ("32d", 256, 256), ("31d", 256, 128), "U", ("22d", 128, 128), ("21d", 128, 64), "U", ("12d", 64, 64), ("11d", 64, None)]


def _pad4(n):
    return (n + 3) // 4 * 4


class SegNet(nn.Module):
    def __init__(self, input_nbr=3, label_nbr=22):
        super().__init__()
        self.input_nbr, self.label_nbr = input_nbr, label_nbr
        for item in _ENC + _DEC:
            if isinstance(item, str):
                continue
            name, cin, cout = item
            cin = input_nbr if cin is None else cin
            cout = label_nbr if cout is None else cout
            setattr(self, "conv" + name, nn.Conv2d(cin, cout, kernel_size=3, padding=1))
            if name != "11d":
                setattr(self, "bn" + name, nn.BatchNorm2d(cout, momentum=0.1))
        self._folded = {}          # name -> (version tag, w [Cout',3,3,Cin'] OHWI, b [Cout'])

    def _layer(self, name):
        conv = getattr(self, "conv" + name)
        bn = getattr(self, "bn" + name, None)
        tensors = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        tag = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._folded.get(name)
        if hit is not None and hit[0] == tag:
            return hit[1], hit[2]
        with torch.no_grad():
            w, b = conv.weight.float(), conv.bias.float()
            if bn is not None:
                scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
                w = w * scale[:, None, None, None]
                b = (b - bn.running_mean.float()) * scale + bn.bias.float()
            cout, cin = w.shape[0], w.shape[1]
            w = w.permute(0, 2, 3, 1)                                     # OIHW -> OHWI
            w = F.pad(w, (0, _pad4(cin) - cin, 0, 0, 0, 0, 0, _pad4(cout) - cout)).contiguous()
            b = F.pad(b, (0, _pad4(cout) - cout)).contiguous()
        self._folded[name] = (tag, w, b)
        return w, b

    def forward(self, x):
        if self.training:
            raise NotImplementedError("SegNet: only the eval() forward runs on the HIP path (training this network is not rebuilt)")
        if not x.is_cuda:
            raise RuntimeError("densefusion_amd needs device tensors (no CPU path): call .cuda() on the input")
        B, C, H, W = x.shape
        if C != self.input_nbr or H % 32 or W % 32:
            raise RuntimeError(f"SegNet.forward: expected [B,{self.input_nbr},H,W] with H, W multiples of 32, got {tuple(x.shape)}")
        L = _lib.lib()
        with torch.no_grad(), _lib.device_guard(x.device):
            a = x.detach().float().permute(0, 2, 3, 1)
            a = F.pad(a, (0, _pad4(C) - C)).contiguous()                  # NHWC, channels padded to 4
            indices = []
            for item in _ENC + _DEC:
                if item == "P":
                    b_, h, w_, c = a.shape
                    y = torch.empty(b_, h // 2, w_ // 2, c, device=a.device)
                    idx = torch.empty(b_, h // 2, w_ // 2, c, dtype=torch.uint8, device=a.device)
                    _lib.check(L.df_maxpool2x2_idx(a.data_ptr(), y.data_ptr(), idx.data_ptr(), b_, h, w_, c, _lib.current_stream()), "maxpool2x2_idx")
                    indices.append(idx)
                    a = y
                elif item == "U":
                    idx = indices.pop()
                    b_, h, w_, c = a.shape
                    y = torch.empty(b_, 2 * h, 2 * w_, c, device=a.device)
                    _lib.check(L.df_maxunpool2x2(a.data_ptr(), idx.data_ptr(), y.data_ptr(), b_, h, w_, c, _lib.current_stream()), "maxunpool2x2")
                    a = y
                else:
                    name = item[0]
                    w, b = self._layer(name)
                    act = 0 if name == "11d" else 1
                    if w.shape[3] >= 256:
                        a = ops.conv3x3_winograd_nhwc(a, w, b, dil=1, act=act)
                    else:
                        a = ops.conv2d_nhwc(a, w, b, stride=1, pad=1, dil=1, act=act)
            return a[..., :self.label_nbr].permute(0, 3, 1, 2).contiguous()
