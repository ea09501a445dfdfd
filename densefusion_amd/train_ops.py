"""Autograd Functions over the native training kernels (csrc/trainops.hip, csrc/igemm.hip): torch.autograd is only
the tape; every forward and backward below is a HIP launch through the C ABI."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from .ops import _desc, _with_splitk, conv2d_nhwc, wgrad


def _st():
    return _lib.current_stream()


def _ck(rc, what):
    _lib.check(rc, what)


# optional per-launch timing of the three MFMA kernels of a training step (bench.py `train` object): events on the launch stream
_PROFILE = None


def profile_begin():
    """Arm per-launch timing of the conv forward / data-gradient / weight-gradient launches (HIP events on the current stream)."""
    global _PROFILE
    _PROFILE = {"fwd": [], "dgrad": [], "wgrad": []}


def profile_end():
    """-> {kind: (milliseconds, FLOPs, launches)} since profile_begin(); synchronises."""
    global _PROFILE
    prof, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    return {k: (sum(a.elapsed_time(b) for a, b, _ in v), sum(f for _, _, f in v), len(v)) for k, v in prof.items()}


class _Timed:
    def __init__(self, kind, flops):
        self.kind, self.flops = kind, flops

    def __enter__(self):
        if _PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _PROFILE is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _PROFILE[self.kind].append((self.a, b, self.flops))
        return False


def _conv_flops(x, w, y):
    return 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]


_SPLITK = {}          # device index -> persistent scratch tensor


class _splitk:
    """Split-K scratch for the convolution launches inside the block: the convolutions of a training pass on small maps (layer3 /
    layer4: a few hundred pixels, reductions of 2304 / 4608) otherwise launch far fewer tiles than the chip has CUs.  The scratch
    travels in every launch's descriptor (``df_conv_desc.splitk_ws``): nothing is registered with the library, the scope is a
    per-thread setting of this module (``ops.current_splitk``), nested scopes are no-ops and an exception inside leaves no state
    behind.  Deterministic (fixed-order reduce); off with DF_TRAIN_NO_SPLITK=1."""

    def __init__(self, device):
        self.dev = device
        self.outer = None

    def __enter__(self):
        from . import ops
        self.outer = ops.current_splitk()
        if self.outer is not None or os.environ.get("DF_TRAIN_NO_SPLITK"):
            return self
        key = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        buf = _SPLITK.get(key)
        if buf is None:
            buf = _SPLITK[key] = torch.empty(64 << 20, dtype=torch.uint8, device=self.dev)
        ops._TLS.splitk = buf
        return self

    def __exit__(self, *a):
        from . import ops
        ops._TLS.splitk = self.outer
        return False


splitk_scope = _splitk


class ConvAct(torch.autograd.Function):
    """y = act(conv(x, w) + bias + res): ONE fused MFMA launch forward; backward = activation gradient kernel, then the
    data-gradient (same MFMA kernel, flipped weights) and weight/bias-gradient MFMA kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, res, slope, stride, pad, dil, act):
        x, w = x.contiguous(), w.contiguous()
        with _Timed("fwd", 0.0) as _t, _splitk(x.device):
            y = conv2d_nhwc(x, w, bias, stride=stride, pad=pad, dil=dil, act=act, res=res.contiguous() if res is not None else None,
                            prelu=slope)
        if _PROFILE is not None and _PROFILE["fwd"]:
            a, b, _ = _PROFILE["fwd"][-1]
            _PROFILE["fwd"][-1] = (a, b, _conv_flops(x, w, y))
        ctx.save_for_backward(x, w, y, slope if slope is not None else torch.empty(0, device=x.device))
        ctx.cfg = (stride, pad, dil, act, bias is not None, res is not None, slope is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, slope = ctx.saved_tensors
        stride, pad, dil, act, has_bias, has_res, has_slope = ctx.cfg
        L = _lib.lib()
        dy = dy.contiguous()
        dslope = None
        with _lib.device_guard(x.device):
            if act:
                g = torch.empty_like(dy)
                partials = None
                if has_slope:
                    dslope = torch.zeros(1, device=x.device)
                    partials = torch.empty(16384, device=x.device)          # DF_ACT_BWD_PARTIALS: per-workgroup sums, added in order
                _ck(L.df_act_bwd(dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), act, slope.data_ptr() if has_slope else None,
                                 dslope.data_ptr() if has_slope else None, partials.data_ptr() if has_slope else None, _st()), "act_bwd")
            else:
                g = dy
            d = _desc(x, w, None, stride, pad, dil)
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                scratch = torch.empty_like(w)
                with _Timed("dgrad", _conv_flops(x, w, y)), _splitk(x.device):
                    _with_splitk(d)
                    _ck(L.df_conv2d_dgrad_nhwc(ctypes.byref(d), g.data_ptr(), dx.data_ptr(), scratch.data_ptr(), 0, _st()), "conv2d_dgrad")
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w)
                db = torch.empty(w.shape[0], device=x.device) if has_bias else None
                with _Timed("wgrad", _conv_flops(x, w, y)):
                    wgrad(d, g, dw, db)
        return dx, dw, db, (g if has_res else None), dslope, None, None, None, None


class MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, C = x.shape
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty(B, OH, OW, C, device=x.device)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_maxpool3s2_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, OH, OW, _st()), "maxpool3s2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_maxpool3s2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, H, W, C, dy.shape[1], dy.shape[2], _st()),
                "maxpool3s2_bwd")
        return dx


class AdaptiveAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty(B, s, s, C, device=x.device)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_adaptive_avgpool(x.data_ptr(), y.data_ptr(), B, H, W, C, s, 0, _st()), "adaptive_avgpool")
        ctx.geom = (B, H, W, C, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, s = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, device=dy.device)
        with _lib.device_guard(dy.device):
            _ck(_lib.lib().df_adaptive_avgpool(dy.data_ptr(), dx.data_ptr(), B, H, W, C, s, 1, _st()), "adaptive_avgpool_bwd")
        return dx, None


class Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW, align):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty(B, OH, OW, C, device=x.device)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_bilinear(x.data_ptr(), y.data_ptr(), B, H, W, C, OH, OW, int(align), 0, _st()), "bilinear")
        ctx.geom = (B, H, W, C, OH, OW, int(align))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, OH, OW, align = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, device=dy.device)
        with _lib.device_guard(dy.device):
            _ck(_lib.lib().df_bilinear(dy.data_ptr(), dx.data_ptr(), B, H, W, C, OH, OW, align, 1, _st()), "bilinear_bwd")
        return dx, None, None, None


class LogSoftmaxLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_logsoftmax(x.data_ptr(), None, y.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], 0, _st()), "logsoftmax")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        with _lib.device_guard(y.device):
            _ck(_lib.lib().df_logsoftmax(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel() // y.shape[-1], y.shape[-1], 1, _st()),
                "logsoftmax_bwd")
        return dx


class Dropout2d(torch.autograd.Function):
    """Dropout2d on [B,H,W,C]: one keep/drop decision per (sample, channel) from a counter-based hash of `seed`."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        B, H, W, C = x.shape
        scale = torch.empty(B, C, device=x.device)
        y = torch.empty_like(x)
        L = _lib.lib()
        with _lib.device_guard(x.device):
            _ck(L.df_dropout2d_mask(scale.data_ptr(), B * C, int(seed) & 0xFFFFFFFF, float(p), _st()), "dropout2d_mask")
            _ck(L.df_channel_scale(x.data_ptr(), scale.data_ptr(), y.data_ptr(), B, H * W, C, _st()), "channel_scale")
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy = dy.contiguous()
        B, H, W, C = dy.shape
        dx = torch.empty_like(dy)
        with _lib.device_guard(dy.device):
            _ck(_lib.lib().df_channel_scale(dy.data_ptr(), scale.data_ptr(), dx.data_ptr(), B, H * W, C, _st()), "channel_scale")
        return dx, None, None


class GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x, idx = x.contiguous(), idx.contiguous()
        n, C = idx.numel(), x.shape[-1]
        y = torch.empty(n, C, device=x.device)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_gather_rows(x.data_ptr(), idx.data_ptr(), y.data_ptr(), n, C, x.shape[0], 0, _st()), "gather_rows")
        ctx.save_for_backward(idx)
        ctx.rows = x.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty(ctx.rows, dy.shape[-1], device=dy.device)
        with _lib.device_guard(dy.device):
            _ck(_lib.lib().df_gather_rows(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), idx.numel(), dy.shape[-1], ctx.rows, 1, _st()),
                "scatter_add_rows")
        return dx, None


class ColMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        rows, C = x.shape
        y = torch.empty(C, device=x.device)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_colmean(x.data_ptr(), y.data_ptr(), rows, C, 0, _st()), "colmean")
        ctx.geom = (rows, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, C = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(rows, C, device=dy.device)
        with _lib.device_guard(dy.device):
            _ck(_lib.lib().df_colmean(dy.data_ptr(), dx.data_ptr(), rows, C, 1, _st()), "colmean_bwd")
        return dx


class Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        with _lib.device_guard(x.device):
            _ck(_lib.lib().df_sigmoid(x.data_ptr(), None, y.data_ptr(), x.numel(), 0, _st()), "sigmoid")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        with _lib.device_guard(y.device):
            _ck(_lib.lib().df_sigmoid(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), 1, _st()), "sigmoid_bwd")
        return dx
