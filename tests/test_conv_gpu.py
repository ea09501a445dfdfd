"""GPU: the implicit-GEMM fp32-MFMA kernel against a plain PyTorch fp32 CPU reference of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, H, W, Cin, Cout, k, stride, pad, dil): every geometry the network uses + ragged edges
GEOMS = [
    (2, 40, 40, 4, 64, 7, 2, 3, 1),       # stem (Cin padded to 4)
    (2, 10, 10, 64, 64, 3, 1, 1, 1),      # layer1
    (1, 10, 10, 64, 128, 3, 2, 1, 1),     # layer2.0.conv1 (stride 2)
    (1, 10, 10, 64, 128, 1, 2, 0, 1),     # layer2.0.downsample
    (3, 5, 5, 256, 256, 3, 1, 2, 2),      # layer3.1 (dilation 2)
    (2, 5, 7, 512, 512, 3, 1, 4, 4),      # layer4.1 (dilation 4, non-square)
    (1, 9, 11, 128, 192, 3, 1, 1, 1),     # Cout not a multiple of the tile, odd sizes
    (16, 1, 1, 2560, 1024, 1, 1, 0, 1),   # psp bottleneck as GEMM rows
    (1, 37, 1, 384, 640, 1, 1, 0, 1),     # per-point GEMM with a ragged M
    (4, 20, 20, 1024, 256, 3, 1, 1, 1),   # up_1: large K = 9216, 128x128 tile path
    (1, 3, 3, 32, 64, 1, 1, 0, 1),        # tiny
]


def _ref(x_nhwc, w_ohwi, bias, stride, pad, dil):
    x = x_nhwc.permute(0, 3, 1, 2).contiguous()
    w = w_ohwi.permute(0, 3, 1, 2).contiguous()
    return F.conv2d(x, w, bias, stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("geom", GEOMS)
def test_conv_matches_torch_cpu(geom):
    from densefusion_amd.ops import conv2d_nhwc
    B, H, W, Cin, Cout, k, s, p, d = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = _ref(x, w, b, s, p, d)
    out = conv2d_nhwc(x.cuda(), w.cuda(), b.cuda(), stride=s, pad=p, dil=d).cpu()
    assert out.shape == ref.shape
    tol = 1e-5 * max(1.0, ref.abs().max().item())
    assert (out - ref).abs().max().item() <= tol


def test_conv_epilogue_residual_relu_prelu_and_channel_offsets():
    from densefusion_amd.ops import conv2d_nhwc
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 6, 96, generator=g)            # use channels 32..95 of a 96-wide row
    w = torch.randn(64, 3, 3, 64, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    res = torch.randn(2, 6, 6, 64, generator=g)
    ref = _ref(x[..., 32:96].contiguous(), w, b, 1, 1, 1)
    relu = torch.relu(ref + res)
    out = torch.full((2, 6, 6, 160), -7.0).cuda()        # write at channel offset 64 of a 160-wide row
    conv2d_nhwc(x.cuda(), w.cuda(), b.cuda(), pad=1, act=1, res=res.cuda(), out=out, out_coff=64, in_coff=32, cin=64)
    out = out.cpu()
    assert (out[..., 64:128] - relu).abs().max().item() < 1e-5 * max(1.0, relu.abs().max().item())
    assert (out[..., :64] == -7.0).all() and (out[..., 128:] == -7.0).all()      # neighbours untouched
    slope = torch.tensor([0.25])
    pre = F.prelu(ref, slope)
    out2 = conv2d_nhwc(x.cuda(), w.cuda(), b.cuda(), pad=1, act=2, prelu=slope.cuda(), in_coff=32, cin=64).cpu()
    assert (out2 - pre).abs().max().item() < 1e-5 * max(1.0, pre.abs().max().item())


def test_conv_is_exact_on_integers():
    """fp32 MFMA is an exact fma chain: small-integer data must come out bit-exact, and an asymmetric
    weight matrix catches a transposed fragment layout."""
    from densefusion_amd.ops import conv2d_nhwc
    g = torch.Generator().manual_seed(9)
    x = torch.randint(-3, 4, (1, 1, 200, 64), generator=g).float()
    w = torch.randint(-3, 4, (192, 1, 1, 64), generator=g).float()
    ref = x.reshape(200, 64) @ w.reshape(192, 64).t()
    out = conv2d_nhwc(x.cuda(), w.cuda()).cpu().reshape(200, 192)
    assert torch.equal(out, ref)


def test_conv_argument_errors():
    from densefusion_amd.ops import conv2d_nhwc
    with pytest.raises(RuntimeError):
        conv2d_nhwc(torch.zeros(1, 4, 4, 6).cuda(), torch.zeros(8, 1, 1, 6).cuda())        # Cin % 4
    with pytest.raises(RuntimeError):
        conv2d_nhwc(torch.zeros(1, 4, 4, 12).cuda(), torch.zeros(8, 3, 3, 12).cuda(), pad=1)  # multi-tap non-pow2


@pytest.mark.parametrize("geom", [
    # (B, H, W, Cin, Cout, dil): trunk shapes (layer3 d1/d2, layer4 d1/d4), odd maps, maps shorter than one tile row
    (2, 20, 20, 256, 256, 1), (2, 20, 20, 256, 256, 2), (1, 20, 20, 512, 512, 4), (2, 15, 20, 256, 512, 1),
    (1, 15, 15, 64, 128, 4), (3, 7, 5, 32, 64, 2), (1, 30, 40, 128, 64, 4), (2, 3, 3, 16, 8, 1), (1, 2, 9, 8, 8, 4),
])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("tile", [2, 4])
def test_winograd_conv_matches_fp64(geom, fused, tile):
    """Winograd F(2x2,3x3) / F(4x4,3x3) path vs an fp64 convolution; the direct kernel is measured beside it: the
    transform-domain result may carry a few times (tile 2) / a few tens of times (tile 4) the direct kernel's rounding error, no more."""
    from densefusion_amd import ops
    import torch.nn.functional as F
    B, H, W, Cin, Cout, dil = geom
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(geom))
    x = (torch.relu(torch.randn(B, H, W, Cin)) * 3).to(dev)
    w = (torch.randn(Cout, 3, 3, Cin) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    bias = torch.randn(Cout, device=dev) if fused else None
    res = torch.randn(B, H, W, Cout, device=dev) if fused else None
    act = 1 if fused else 0
    want = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double() if fused else None, 1, dil, dil)
    want = want.permute(0, 2, 3, 1)
    if fused:
        want = torch.relu(want + res.double())
    got_w = ops.conv3x3_winograd_nhwc(x, w, bias, dil=dil, act=act, res=res, tile=tile)
    got_d = ops.conv2d_nhwc(x, w, bias, stride=1, pad=dil, dil=dil, act=act, res=res)
    scale = float(want.abs().max())
    err_w = float((got_w.double() - want).abs().max()) / scale
    err_d = float((got_d.double() - want).abs().max()) / scale
    assert err_w < (3e-6 if tile == 2 else 2e-5), (err_w, err_d)
    assert err_w < (8 if tile == 2 else 60) * err_d + 1e-7, (err_w, err_d)


def test_winograd_rejects_other_geometries():
    from densefusion_amd import ops
    dev = torch.device("cuda:0")
    x = torch.randn(1, 8, 8, 8, device=dev)
    w5 = torch.randn(8, 5, 5, 8, device=dev)
    with pytest.raises(RuntimeError):
        ops.conv3x3_winograd_nhwc(x, w5)


def test_random_geometries_direct_and_winograd():
    """Seeded sweep over geometries the fixed lists do not hit: ragged M / Cout against both tile sizes, strides, dilations,
    residual + bias + activations, weight operands large enough for the column-tile groups, and Winograd maps of arbitrary
    (odd, tiny, non-square) size and dilation 1..4 -- each against an fp64 convolution."""
    from densefusion_amd import ops
    import random
    rnd = random.Random(1234)
    dev = torch.device("cuda:0")
    cases = []
    for _ in range(28):
        k = rnd.choice([1, 1, 3, 3, 7])
        cin = rnd.choice([4, 8, 16, 32, 64, 128, 256]) if k > 1 else 4 * rnd.randint(1, 96)
        cout = 4 * rnd.randint(1, 80)
        stride = rnd.choice([1, 1, 2]) if k > 1 else rnd.choice([1, 2])
        dil = rnd.choice([1, 2, 4]) if k == 3 else 1
        pad = dil * (k // 2) if rnd.random() < 0.8 else 0
        cases.append((rnd.randint(1, 3), rnd.randint(1 + dil * (k - 1), 33), rnd.randint(1 + dil * (k - 1), 29), cin, cout, k, stride, pad, dil))
    cases += [(1, 40, 25, 1024, 1536, 1, 1, 0, 1), (2, 9, 11, 512, 1028, 3, 1, 1, 1)]      # > 3 MB of weights: column-tile groups
    for (B, H, W, Cin, Cout, k, s, p, d) in cases:
        g = torch.Generator().manual_seed(B * 1000003 + H * 1009 + W * 13 + Cin + Cout + k)
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        act = (B + H) % 3
        slope = torch.tensor([0.25], device=dev) if act == 2 else None
        want = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), s, p, d).permute(0, 2, 3, 1)
        res = torch.randn(*want.shape, generator=g).to(dev) if (H + W) % 2 else None
        if res is not None:
            want = want + res.double()
        want = torch.relu(want) if act == 1 else (torch.where(want > 0, want, want * 0.25) if act == 2 else want)
        got = ops.conv2d_nhwc(x, w, b, stride=s, pad=p, dil=d, act=act, res=res, prelu=slope)
        err = float((got.double() - want).abs().max()) / max(float(want.abs().max()), 1e-9)
        assert err < 2e-5, ((B, H, W, Cin, Cout, k, s, p, d), err)
    for _ in range(16):
        B, H, W, d = rnd.randint(1, 3), rnd.randint(1, 31), rnd.randint(1, 37), rnd.randint(1, 4)
        Cin, Cout = 4 * rnd.randint(1, 40), 4 * rnd.randint(1, 40)
        g = torch.Generator().manual_seed(H * 131 + W * 17 + d + Cin)
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
        want = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, 1, d, d).permute(0, 2, 3, 1)
        for tile in (2, 4):
            got = ops.conv3x3_winograd_nhwc(x, w, dil=d, tile=tile)
            err = float((got.double() - want).abs().max()) / max(float(want.abs().max()), 1e-9)
            assert err < (1e-5 if tile == 2 else 4e-5), ((B, H, W, Cin, Cout, d, tile), err)


@pytest.mark.parametrize("geom", [(2, 40, 56, 4, 64, 7, 2, 3, 1), (2, 20, 28, 64, 128, 3, 2, 1, 1), (1, 21, 17, 64, 128, 1, 2, 0, 1),
                                  (2, 12, 12, 128, 63, 1, 1, 0, 1), (1, 10, 14, 128, 21, 1, 1, 0, 1)])
def test_v1_kernel_paths(geom):
    """What the un-pipelined kernel (igemm_f32_kernel) is kept for, against torch: the data gradient of the strided
    convolutions (stem 7x7/2, layer2's 3x3/2 and 1x1/2: the forward kernel run with input dilation `up` = stride) and output
    widths that are not a multiple of 4 (the 63 / 21-wide last head layers of the training graph: whole launch on v1)."""
    import ctypes
    from densefusion_amd import _lib
    from densefusion_amd.ops import _desc, conv2d_nhwc
    B, H, W, Cin, Cout, k, s, p, d = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).cuda()
    if Cout % 4:
        b = torch.randn(Cout, generator=g).cuda()
        got = conv2d_nhwc(x, w, b, stride=s, pad=p, dil=d, act=1)
        want = torch.relu(_ref(x.cpu(), w.cpu(), b.cpu(), s, p, d))
        assert (got.cpu() - want).abs().max() < 2e-5
        return
    xr, wr = x.clone().requires_grad_(), w.clone()
    y = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), stride=s, padding=p, dilation=d).permute(0, 2, 3, 1)
    dy = torch.randn(y.shape, generator=g).cuda()
    y.backward(dy)
    x, w = x.contiguous(), w.contiguous()
    dx = torch.empty_like(x)
    scratch = torch.empty_like(w)
    dsc = _desc(x, w, None, s, p, d)
    _lib.check(_lib.lib().df_conv2d_dgrad_nhwc(ctypes.byref(dsc), dy.contiguous().data_ptr(), dx.data_ptr(), scratch.data_ptr(), 0,
                                                _lib.current_stream()), "conv2d_dgrad")
    assert (dx - xr.grad).abs().max() < 2e-5 * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [
    # (B, H, W, Cin, Cout, k, pad): row counts that end inside a 128-row tile, for the plain-GEMM, tap-uniform and general loaders,
    # with column counts that end inside a column tile
    (1, 25, 30, 64, 128, 1, 0), (3, 17, 19, 256, 576, 1, 0), (2, 15, 15, 64, 64, 3, 1), (1, 9, 11, 32, 192, 3, 1), (2, 13, 7, 4, 64, 7, 3),
    (5, 100, 37, 64, 320, 1, 0),
])
def test_conv_writes_nothing_outside_its_output(geom):
    """The epilogue drops rows past M and columns past Cout through the buffer descriptor's range (voffset + soffset against
    num_records -- tools/dev/soffset_check.hip shows the sum is what the hardware checks): the output is a window of a larger
    tensor filled with a sentinel, which must survive around it."""
    from densefusion_amd.ops import conv2d_nhwc
    B, H, W, Cin, Cout, k, pad = geom
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(geom))
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    ld, coff, guard_rows = Cout + 8, 4, 160
    rows = B * H * W
    big = torch.full((rows + guard_rows, ld), 7.25, device=dev)
    out = big[:rows].view(B, H, W, ld)
    conv2d_nhwc(x, w, None, stride=1, pad=pad, act=1, out=out, out_coff=coff)
    torch.cuda.synchronize()
    assert bool((big[rows:] == 7.25).all()), "rows past M were written"
    assert bool((big[:rows, :coff] == 7.25).all()) and bool((big[:rows, coff + Cout:] == 7.25).all()), "columns outside the channel window were written"
    want = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, 1, pad)).permute(0, 2, 3, 1)
    got = out[..., coff:coff + Cout].double()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


_LOADER_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from densefusion_amd.ops import conv2d_nhwc
dev = torch.device("cuda:0")
outs = []
for i, (B, H, W, Cin, Cout, k, stride, pad, dil) in enumerate([
        (40, 32, 32, 64, 128, 1, 1, 0, 1),      # plain GEMM (single tap), full grid
        (16, 40, 40, 64, 64, 3, 1, 1, 1),       # tap-uniform loader, 3x3
        (16, 40, 40, 64, 128, 3, 2, 1, 1),      # tap-uniform, stride 2
        (8, 20, 20, 256, 512, 3, 1, 4, 4),      # tap-uniform, dilation 4
        (3, 9, 11, 32, 64, 3, 1, 1, 1)]):       # small grid (software-pipelined kernel either way)
    torch.manual_seed(100 + i)
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    outs.append(conv2d_nhwc(x, w, b, stride=stride, pad=pad, dil=dil, act=1).cpu())
torch.save(outs, sys.argv[2])
"""


@pytest.mark.gpu
def test_specialised_loaders_are_bit_identical_to_the_general_one(tmp_path):
    """The plain-GEMM and tap-uniform loaders only change HOW a k tile's addresses are formed: every output element still adds the
    same products in the same order, so results must equal the general loader's bit for bit (two child processes: the dev
    switch DF_IGEMM_NOPURE is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for tag, env in (("special", {}), ("general", {"DF_DEV_LIB": "1", "DF_IGEMM_NOPURE": "1"})):
        f = str(tmp_path / f"{tag}.pt")
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", _LOADER_SCRIPT, root, f], check=True, env=e, timeout=300)
        files.append(f)
    a, b = torch.load(files[0]), torch.load(files[1])
    assert len(a) == len(b) == 5
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), f"case {i}: loaders disagree, max diff {float((u - v).abs().max())}"


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [
    # (B, H, W, Cin, Cout, k, pad, dil): the layer3 / layer4 shapes of an 8-frame training pass (few tiles, long reductions) and a 1x1
    (8, 20, 20, 512, 512, 3, 4, 4), (8, 20, 20, 256, 256, 3, 2, 2), (4, 15, 20, 256, 512, 3, 4, 4), (8, 20, 20, 2048, 256, 1, 0, 1),
    (1, 20, 20, 512, 512, 3, 4, 4),
])
def test_splitk_convolution(geom):
    """Opt-in split-K (df_conv_desc.splitk_ws, what the training ops pass): against an fp64 convolution, against the unsplit launch
    (fp32 re-association only), bit-reproducible run to run, residual + bias + ReLU applied once by the reduce kernel -- and the data
    gradient through the same path."""
    import ctypes
    from densefusion_amd import _lib, train_ops
    from densefusion_amd.ops import conv2d_nhwc, _desc
    B, H, W, Cin, Cout, k, pad, dil = geom
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(geom))
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) * (1.0 / (k * k * Cin)) ** 0.5
    bias = torch.randn(Cout, device=dev)
    res = torch.randn(B, H, W, Cout, device=dev)
    want = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), bias.double(), 1, pad, dil)
                      .permute(0, 2, 3, 1) + res.double())
    plain = conv2d_nhwc(x, w, bias, stride=1, pad=pad, dil=dil, act=1, res=res)
    assert _lib.lib().df_conv_last_splitk() == 1
    with train_ops._splitk(dev):
        a = conv2d_nhwc(x, w, bias, stride=1, pad=pad, dil=dil, act=1, res=res)
        assert _lib.lib().df_conv_last_splitk() > 1, "this geometry is meant to be split"
        with train_ops._splitk(dev):                      # nested scope: same scratch, still split
            b = conv2d_nhwc(x, w, bias, stride=1, pad=pad, dil=dil, act=1, res=res)
        # PReLU through the reduce kernel, into a wider output at a channel offset
        slope = torch.tensor([0.2], device=dev)
        wide = torch.zeros(B, H, W, Cout + 8, device=dev)
        conv2d_nhwc(x, w, bias, stride=1, pad=pad, dil=dil, act=2, prelu=slope, out=wide, out_coff=8)
        assert _lib.lib().df_conv_last_splitk() > 1
    from densefusion_amd import ops as _ops
    assert _ops.current_splitk() is None                  # the scope leaves nothing behind
    assert torch.equal(a, b), "split-K is not reproducible run to run"
    pre = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), bias.double(), 1, pad, dil).permute(0, 2, 3, 1)
    want_p = torch.where(pre > 0, pre, 0.2 * pre)
    assert float((wide[..., 8:].double() - want_p).abs().max()) <= 2e-5 * max(1.0, float(want_p.abs().max())) and float(wide[..., :8].abs().max()) == 0.0
    scale = max(1.0, float(want.abs().max()))
    assert float((a.double() - want).abs().max()) <= 2e-5 * scale
    assert float((a - plain).abs().max()) <= 2e-5 * scale
    # data gradient: dx = conv_transpose(dy, w)
    dy = torch.randn(B, H, W, Cout, device=dev)
    d = _desc(x, w, None, 1, pad, dil)
    L = _lib.lib()
    outs = []
    for use in (False, True, True):
        dx = torch.empty_like(x)
        scratch = torch.empty_like(w)
        if use:
            with train_ops._splitk(dev):
                d = _desc(x, w, None, 1, pad, dil)       # (the descriptor carries the scope's scratch)
                _lib.check(L.df_conv2d_dgrad_nhwc(ctypes.byref(d), dy.data_ptr(), dx.data_ptr(), scratch.data_ptr(), 0, _lib.current_stream()), "dgrad")
                assert L.df_conv_last_splitk() > 1 or k == 1          # (the 1x1 case's data gradient has 400 tiles and K = 256: not split)
        else:
            d = _desc(x, w, None, 1, pad, dil)
            _lib.check(L.df_conv2d_dgrad_nhwc(ctypes.byref(d), dy.data_ptr(), dx.data_ptr(), scratch.data_ptr(), 0, _lib.current_stream()), "dgrad")
        outs.append(dx)
    # accumulate = 1 through the reduce kernel (res == out)
    with train_ops._splitk(dev):
        d = _desc(x, w, None, 1, pad, dil)
        acc = outs[1].clone()
        _lib.check(L.df_conv2d_dgrad_nhwc(ctypes.byref(d), dy.data_ptr(), acc.data_ptr(), torch.empty_like(w).data_ptr(), 1, _lib.current_stream()), "dgrad")
    assert float((acc - 2 * outs[1]).abs().max()) <= 1e-5 * max(1.0, float(outs[1].abs().max()))
    want_dx = torch.nn.grad.conv2d_input((B, Cin, H, W), w.permute(0, 3, 1, 2).double(), dy.permute(0, 3, 1, 2).double(), 1, pad, dil).permute(0, 2, 3, 1)
    assert torch.equal(outs[1], outs[2])
    s2 = max(1.0, float(want_dx.abs().max()))
    assert float((outs[1].double() - want_dx).abs().max()) <= 3e-5 * s2
    assert float((outs[0].double() - want_dx).abs().max()) <= 3e-5 * s2


def _multi_desc(x, w, out, stride, pad, dil, act=0, bias=None, res=None):
    import ctypes
    from densefusion_amd import _lib
    d = _lib.ConvDesc()
    d.in_, d.wgt, d.out = x.data_ptr(), w.data_ptr(), (out.data_ptr() if out is not None else None)
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.Cin, d.in_ld, d.in_coff = w.shape[-1], x.shape[-1], 0
    d.Cout, d.out_ld, d.out_coff = w.shape[0], w.shape[0], 0
    d.res_ld, d.res_coff = (res.shape[-1] if res is not None else 0), 0
    d.KH, d.KW, d.stride, d.pad, d.dil, d.act = w.shape[1], w.shape[2], stride, pad, dil, act
    return d


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,act,use_res", [(64, 64, 3, 1, 1, 1, 1, True), (64, 128, 3, 2, 1, 1, 1, False), (64, 128, 1, 2, 0, 1, 0, False),
                                                                   (4, 64, 7, 2, 3, 1, 1, False), (512, 512, 3, 1, 4, 4, 1, True), (128, 128, 3, 1, 2, 2, 0, False)])
def test_multi_bucket_convolution_is_bit_identical_to_per_bucket_launches(cin, cout, k, stride, pad, dil, act, use_res):
    """df_conv2d_nhwc_multi (csrc/igemm.hip igemm_f32_v4_multi_kernel): several crop-size buckets -- different map sizes and batch
    counts, pixel rows concatenated -- in ONE launch equal one df_conv2d_nhwc launch per bucket (no split-K) bit for bit; and the
    multi-bucket weight gradient equals the fp64 reference of the summed per-bucket gradients."""
    import ctypes
    from densefusion_amd import _lib
    from densefusion_amd.ops import conv2d_nhwc
    L = _lib.lib()
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    sizes = [(2, 10, 20), (1, 40, 40), (1, 20, 20), (3, 7, 9), (1, 30, 40)]
    if cin >= 512:
        sizes = [(2, 5, 10), (1, 20, 20), (1, 10, 10), (1, 15, 20)]
    w = (torch.randn(cout, k, k, cin, generator=g) * 0.05).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    xs = [torch.randn(b, h, wd, cin, generator=g).cuda() for b, h, wd in sizes]
    oh = lambda v: (v + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ress = [torch.randn(b, oh(h), oh(wd), cout, generator=g).cuda() for b, h, wd in sizes] if use_res else [None] * len(sizes)
    want = [conv2d_nhwc(x, w, bias, stride=stride, pad=pad, dil=dil, act=act, res=r) for x, r in zip(xs, ress)]        # ops.conv2d_nhwc passes no split-K scratch here
    xcat = torch.cat([x.reshape(-1, cin) for x in xs]).contiguous()
    rcat = torch.cat([r.reshape(-1, cout) for r in ress]).contiguous() if use_res else None
    rows_out = sum(t.shape[0] * t.shape[1] * t.shape[2] for t in want)
    out = torch.full((rows_out, cout), float("nan"), device="cuda")
    d = _multi_desc(xcat, w, out, stride, pad, dil, act, bias, rcat)
    arr = ctypes.c_int * len(sizes)
    cB, cH, cW = arr(*[s[0] for s in sizes]), arr(*[s[1] for s in sizes]), arr(*[s[2] for s in sizes])
    _lib.check(L.df_conv2d_nhwc_multi(ctypes.byref(d), len(sizes), cB, cH, cW, _lib.current_stream()), "conv2d_nhwc_multi")
    got = out.cpu()
    ref = torch.cat([t.reshape(-1, cout) for t in want]).cpu()
    assert torch.equal(got, ref), f"max diff {(got - ref).abs().max().item():.3e}"
    # weight gradient over all buckets in one contraction vs fp64
    dy = torch.randn(rows_out, cout, generator=g).cuda()
    dw, db = torch.empty_like(w), torch.empty(cout, device="cuda")
    d.out = None
    need = L.df_conv2d_wgrad_multi_workspace_bytes(ctypes.byref(d), len(sizes), cB, cH, cW)
    ws = torch.empty(max(int(need), 4), dtype=torch.uint8, device="cuda")
    _lib.check(L.df_conv2d_wgrad_nhwc_multi(ctypes.byref(d), len(sizes), cB, cH, cW, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _lib.current_stream()), "conv2d_wgrad_multi")
    dw_ref = torch.zeros(cout, cin, k, k, dtype=torch.float64)
    r0 = 0
    for x, (b, h, wd) in zip(xs, sizes):
        n = b * oh(h) * oh(wd)
        gy = dy[r0:r0 + n].reshape(b, oh(h), oh(wd), cout).permute(0, 3, 1, 2).double().cpu()
        xin = x.permute(0, 3, 1, 2).double().cpu()
        dw_ref += torch.nn.grad.conv2d_weight(xin, (cout, cin, k, k), gy, stride=stride, padding=pad, dilation=dil)
        r0 += n
    dw_ref = dw_ref.permute(0, 2, 3, 1)
    scale = dw_ref.abs().max().item()
    assert (dw.cpu().double() - dw_ref).abs().max().item() <= 2e-5 * scale
    db_ref = dy.double().sum(0).cpu()
    assert (db.cpu().double() - db_ref).abs().max().item() <= 2e-5 * db_ref.abs().max().item()
