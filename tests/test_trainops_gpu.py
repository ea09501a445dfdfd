"""Native training kernels (csrc/trainops.hip) forward + backward against stock fp32 torch ops of the same layer
(the layers the reference's training graph uses: lib/extractors.py:114-124, lib/pspnet.py:13-77, lib/network.py:95-132)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _pair(fn_native, fn_torch, x, tol=1e-5, exact_fwd=False):
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya, yb = fn_native(xa), fn_torch(xb)
    assert ya.shape == yb.shape
    if exact_fwd:
        assert torch.equal(ya, yb)
    else:
        torch.testing.assert_close(ya, yb, rtol=tol, atol=tol)
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=tol, atol=tol)


@pytest.mark.parametrize("shape", [(1, 40, 40, 64), (2, 33, 47, 8)])
def test_maxpool(shape):
    from densefusion_amd import train_ops as T
    dev = _dev()
    torch.manual_seed(0)
    x = torch.randn(*shape, device=dev)
    _pair(T.MaxPool3s2.apply, lambda t: _nhwc(F.max_pool2d(_nchw(t), 3, 2, 1)), x, exact_fwd=True)
    # ties: quantised input so several window entries share the maximum -> the FIRST one takes the gradient
    xq = torch.randint(0, 3, shape, device=dev).float()
    _pair(T.MaxPool3s2.apply, lambda t: _nhwc(F.max_pool2d(_nchw(t), 3, 2, 1)), xq, exact_fwd=True)


@pytest.mark.parametrize("hw", [(20, 20), (10, 15), (7, 5)])
@pytest.mark.parametrize("s", [1, 2, 3, 6])
def test_adaptive_avgpool(hw, s):
    from densefusion_amd import train_ops as T
    dev = _dev()
    torch.manual_seed(1)
    x = torch.randn(2, hw[0], hw[1], 16, device=dev)
    _pair(lambda t: T.AdaptiveAvgPool.apply(t, s), lambda t: _nhwc(F.adaptive_avg_pool2d(_nchw(t), (s, s))), x)


@pytest.mark.parametrize("case", [((6, 6), (20, 20), False), ((1, 1), (10, 15), False), ((3, 3), (7, 5), False),
                                  ((10, 10), (20, 20), True), ((5, 8), (10, 16), True)])
def test_bilinear(case):
    from densefusion_amd import train_ops as T
    (h, w), (oh, ow), align = case
    dev = _dev()
    torch.manual_seed(2)
    x = torch.randn(2, h, w, 12, device=dev)
    # (tolerance: the input gradient sums up to oh*ow = 150 terms per element and torch's backward adds them with atomics in
    # an order that changes from run to run -- 1.6e-5 apart was seen on the 1x1 -> 10x15 case)
    _pair(lambda t: T.Bilinear.apply(t, oh, ow, align),
          lambda t: _nhwc(F.interpolate(_nchw(t), size=(oh, ow), mode="bilinear", align_corners=align)), x, tol=5e-5)


def test_logsoftmax_sigmoid_colmean_gather():
    from densefusion_amd import train_ops as T
    dev = _dev()
    torch.manual_seed(3)
    x = torch.randn(1, 24, 24, 32, device=dev) * 3
    _pair(T.LogSoftmaxLast.apply, lambda t: F.log_softmax(t, dim=3), x)
    s = torch.randn(500, 1, device=dev) * 4
    _pair(T.Sigmoid.apply, torch.sigmoid, s, tol=1e-6)
    m = torch.randn(500, 1024, device=dev)
    _pair(T.ColMean.apply, lambda t: t.mean(dim=0), m)
    rows = torch.randn(24 * 24, 32, device=dev)
    idx = torch.randint(0, 24 * 24, (500,), device=dev)              # with repeats: backward must accumulate
    _pair(lambda t: T.GatherRows.apply(t, idx), lambda t: t[idx], rows, exact_fwd=True)


def test_dropout2d():
    from densefusion_amd import train_ops as T
    dev = _dev()
    x = torch.ones(4, 8, 8, 1024, device=dev, requires_grad=True)
    for p in (0.3, 0.15):
        y = T.Dropout2d.apply(x, p, 1234)
        per = y.detach()[:, 0, 0, :]
        assert torch.equal(y, per[:, None, None, :].expand_as(y))    # whole channels kept or dropped
        vals = per.unique()
        assert vals.numel() == 2 and vals[0] == 0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
        drop = float((per == 0).float().mean())
        assert abs(drop - p) < 0.03
        assert torch.equal(T.Dropout2d.apply(x, p, 1234), y)         # same seed -> same mask
        assert not torch.equal(T.Dropout2d.apply(x, p, 1235), y)
        (g,) = torch.autograd.grad(y.sum(), x)
        assert torch.equal(g, y)                                       # d/dx = the same per-channel scale


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_act(act, with_res):
    from densefusion_amd import train_ops as T
    dev = _dev()
    torch.manual_seed(4)
    B, H, W, Cin, Cout = 1, 20, 20, 64, 128
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    res = torch.randn(B, H, W, Cout, device=dev) if with_res else None
    slope = torch.tensor([0.25], device=dev) if act == 2 else None

    leaves_a = [t.clone().requires_grad_(True) if t is not None else None for t in (x, w, b, res, slope)]
    leaves_b = [t.clone().requires_grad_(True) if t is not None else None for t in (x, w, b, res, slope)]
    ya = T.ConvAct.apply(*leaves_a, 1, 1, 1, act)
    xb, wb, bb, rb, sb = leaves_b
    yb = F.conv2d(_nchw(xb), wb.permute(0, 3, 1, 2), bb, 1, 1, 1)
    if rb is not None:
        yb = yb + _nchw(rb)
    yb = F.relu(yb) if act == 1 else (F.prelu(yb, sb) if act == 2 else yb)
    yb = _nhwc(yb)
    torch.testing.assert_close(ya, yb, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    for la, lb in zip(leaves_a, leaves_b):
        if la is not None:
            scale = float(lb.grad.abs().max()) + 1e-6
            assert float((la.grad - lb.grad).abs().max()) <= 2e-4 * scale + 1e-5
