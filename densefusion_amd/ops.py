"""Single-layer entry points over the C ABI (unit parity tests, ad-hoc use)."""
from __future__ import annotations

import ctypes
import threading

import torch

from . import _lib


_TLS = threading.local()        # .splitk: the split-K scratch tensor of the enclosing train_ops.splitk_scope (per thread), or None


def current_splitk():
    return getattr(_TLS, "splitk", None)


def _with_splitk(d):
    buf = current_splitk()
    if buf is not None:
        d.splitk_ws, d.splitk_ws_bytes = buf.data_ptr(), buf.numel()
    return d


def conv2d_nhwc(x, w, bias=None, stride=1, pad=0, dil=1, act=0, res=None, prelu=None, out=None, out_coff=0,
                in_coff=0, cin=None):
    """Channels-last convolution on the fp32 matrix cores.

    x [B,H,W,ld] (channels [in_coff, in_coff+cin) used); w [Cout,KH,KW,Cin]; returns out [B,OH,OW,out_ld].
    """
    B, H, W, in_ld = x.shape
    Cout, KH, KW, Cin = w.shape
    cin = Cin if cin is None else cin
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty(B, OH, OW, Cout, device=x.device, dtype=torch.float32)
    d = _with_splitk(_lib.ConvDesc())
    d.in_, d.wgt, d.out = _lib.dptr(x), _lib.dptr(w), _lib.dptr(out)
    d.bias = _lib.dptr(bias) if bias is not None else None
    d.res = _lib.dptr(res) if res is not None else None
    d.prelu = _lib.dptr(prelu) if prelu is not None else None
    d.B, d.H, d.W, d.Cin, d.in_ld, d.in_coff = B, H, W, cin, in_ld, in_coff
    d.OH, d.OW, d.Cout, d.out_ld, d.out_coff = OH, OW, Cout, out.shape[-1], out_coff
    d.res_ld, d.res_coff = (res.shape[-1] if res is not None else 0), 0
    d.KH, d.KW, d.stride, d.pad, d.dil, d.act = KH, KW, stride, pad, dil, act
    with _lib.device_guard(x.device):
        _lib.check(_lib.lib().df_conv2d_nhwc(ctypes.byref(d), _lib.current_stream()), "conv2d_nhwc")
    return out


def _desc(x, w, out, stride, pad, dil, act=0, bias=None):
    B, H, W, in_ld = x.shape
    Cout, KH, KW, Cin = w.shape
    d = _with_splitk(_lib.ConvDesc())
    d.in_, d.wgt, d.out = _lib.dptr(x), _lib.dptr(w), (_lib.dptr(out) if out is not None else None)
    d.bias = _lib.dptr(bias) if bias is not None else None
    d.B, d.H, d.W, d.Cin, d.in_ld, d.in_coff = B, H, W, Cin, in_ld, 0
    d.OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    d.OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    d.Cout, d.out_ld, d.out_coff = Cout, Cout, 0
    d.KH, d.KW, d.stride, d.pad, d.dil, d.act = KH, KW, stride, pad, dil, act
    return d


class ConvNHWC(torch.autograd.Function):
    """y = conv(x [B,H,W,Cin], w [Cout,KH,KW,Cin]) + bias on the fp32-MFMA kernels, with native data and
    weight gradients (df_conv2d_dgrad_nhwc / df_conv2d_wgrad_nhwc).  The activation is left to the caller."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil):
        x, w = x.contiguous(), w.contiguous()
        y = conv2d_nhwc(x, w, bias, stride=stride, pad=pad, dil=dil)
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad, dil, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil, has_bias = ctx.geom
        dy = dy.contiguous()
        L = _lib.lib()
        d = _desc(x, w, None, stride, pad, dil)
        dx = dw = db = None
        with _lib.device_guard(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                scratch = torch.empty_like(w)
                _lib.check(L.df_conv2d_dgrad_nhwc(ctypes.byref(d), dy.data_ptr(), dx.data_ptr(), scratch.data_ptr(), 0,
                                                  _lib.current_stream()), "conv2d_dgrad")
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w)
                db = torch.empty(w.shape[0], device=x.device) if has_bias else None
                wgrad(d, dy, dw, db)
        return dx, dw, db, None, None, None


def wgrad(d, dy, dw, db):
    """``df_conv2d_wgrad_nhwc`` with its split-pixel workspace (deterministic two-pass reduction)."""
    L = _lib.lib()
    need = L.df_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(int(need), 4), dtype=torch.uint8, device=dy.device)
    _lib.check(L.df_conv2d_wgrad_nhwc(ctypes.byref(d), dy.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                      ws.data_ptr(), ws.numel(), _lib.current_stream()), "conv2d_wgrad")


def conv3x3_winograd_nhwc(x, w, bias=None, dil=1, act=0, res=None, tile=2):
    """3x3 / stride 1 / pad == dil convolution through the Winograd F(tile x tile, 3x3) domain, tile 2 or 4
    (``df_conv3x3_winograd_tile_nhwc``).  x [B,H,W,Cin], w [Cout,3,3,Cin] -> [B,H,W,Cout]."""
    B, H, W, in_ld = x.shape
    Cout, KH, KW, Cin = w.shape
    out = torch.empty(B, H, W, Cout, device=x.device, dtype=torch.float32)
    d = _lib.ConvDesc()
    d.in_, d.wgt, d.out = _lib.dptr(x), _lib.dptr(w), _lib.dptr(out)
    d.bias = _lib.dptr(bias) if bias is not None else None
    d.res = _lib.dptr(res) if res is not None else None
    d.prelu = None
    d.B, d.H, d.W, d.Cin, d.in_ld, d.in_coff = B, H, W, Cin, in_ld, 0
    d.OH, d.OW, d.Cout, d.out_ld, d.out_coff = H, W, Cout, Cout, 0
    d.res_ld, d.res_coff = (res.shape[-1] if res is not None else 0), 0
    d.KH, d.KW, d.stride, d.pad, d.dil, d.act = KH, KW, 1, dil, dil, act
    L = _lib.lib()
    with _lib.device_guard(x.device):
        need = L.df_conv3x3_winograd_tile_scratch_bytes(ctypes.byref(d), tile)
        if need == 0:
            _lib.check(-1, "conv3x3_winograd_scratch_bytes")
        scratch = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        _lib.check(L.df_conv3x3_winograd_tile_nhwc(ctypes.byref(d), tile, scratch.data_ptr(), need, _lib.current_stream()), "conv3x3_winograd_nhwc")
    return out
