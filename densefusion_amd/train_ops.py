"""Autograd Functions over the native training kernels (csrc/trainops.hip, csrc/igemm.hip): torch.autograd is only
the tape; every forward and backward below is a HIP launch through the C ABI."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .ops import _desc, conv2d_nhwc


def _st():
    return _lib.current_stream()


def _ck(rc, what):
    _lib.check(rc, what)


class ConvAct(torch.autograd.Function):
    """y = act(conv(x, w) + bias + res): ONE fused MFMA launch forward; backward = activation gradient kernel, then the
    data-gradient (same MFMA kernel, flipped weights) and weight/bias-gradient MFMA kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, res, slope, stride, pad, dil, act):
        x, w = x.contiguous(), w.contiguous()
        y = conv2d_nhwc(x, w, bias, stride=stride, pad=pad, dil=dil, act=act, res=res.contiguous() if res is not None else None,
                        prelu=slope)
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
                if has_slope:
                    dslope = torch.zeros(1, device=x.device)
                _ck(L.df_act_bwd(dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), act, slope.data_ptr() if has_slope else None,
                                 dslope.data_ptr() if has_slope else None, _st()), "act_bwd")
            else:
                g = dy
            d = _desc(x, w, None, stride, pad, dil)
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                scratch = torch.empty_like(w)
                _ck(L.df_conv2d_dgrad_nhwc(ctypes.byref(d), g.data_ptr(), dx.data_ptr(), scratch.data_ptr(), 0, _st()), "conv2d_dgrad")
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w)
                db = torch.empty(w.shape[0], device=x.device) if has_bias else None
                _ck(L.df_conv2d_wgrad_nhwc(ctypes.byref(d), g.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, _st()),
                    "conv2d_wgrad")
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
