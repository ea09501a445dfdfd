"""GPU: data / weight / bias gradients of the MFMA convolution against torch CPU autograd of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, H, W, Cin, Cout, k, stride, pad, dil)
GEOMS = [
    (2, 10, 10, 64, 64, 3, 1, 1, 1),      # layer1
    (2, 10, 10, 64, 128, 3, 2, 1, 1),     # layer2.0.conv1 (stride 2 -> input-dilated dgrad)
    (2, 10, 10, 64, 128, 1, 2, 0, 1),     # layer2.0.downsample
    (2, 5, 7, 256, 256, 3, 1, 2, 2),      # dilation 2
    (1, 6, 6, 512, 512, 3, 1, 4, 4),      # dilation 4
    (3, 1, 1, 384, 640, 1, 1, 0, 1),      # per-point GEMM rows (B = points)
    (700, 1, 1, 32, 64, 1, 1, 0, 1),      # e_conv1 over many points
    (1, 20, 20, 1024, 256, 3, 1, 1, 1),   # up_1
    (2, 9, 11, 64, 256, 3, 1, 1, 1),      # odd sizes (multi-tap dgrad needs a power-of-two Cout)
]


@pytest.mark.parametrize("geom", GEOMS)
def test_conv_gradients_match_torch_cpu(geom):
    from densefusion_amd.ops import ConvNHWC
    B, H, W, Cin, Cout, k, s, p, d = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    xc, wc, bc = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yc = F.conv2d(xc.permute(0, 3, 1, 2), wc.permute(0, 3, 1, 2), bc, stride=s, padding=p, dilation=d).permute(0, 2, 3, 1)
    gy = torch.randn(yc.shape, generator=g)
    yc.backward(gy)
    xg, wg, bg = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    yg = ConvNHWC.apply(xg, wg, bg, s, p, d)
    yg.backward(gy.cuda())
    for got, want, name in ((yg, yc, "y"), (xg.grad, xc.grad, "dx"), (wg.grad, wc.grad, "dw"), (bg.grad, bc.grad, "db")):
        err = (got.detach().cpu() - want.detach()).abs().max().item()
        tol = 2e-5 * max(1.0, want.abs().max().item())
        assert err <= tol, (name, err, tol)
