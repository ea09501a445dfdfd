"""GPU: the SegNet mirror (BatchNorm folded, fp32-MFMA convs incl. the Winograd-domain ones, 2x2 pool / un-pool kernels)
against the golden of the imported reference module and against the CPU restatement on a full 480 x 640 frame."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from densefusion_amd import synth
from oracle import segnet_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _net(seed):
    from densefusion_amd.vanilla_segmentation.segnet import SegNet
    net = SegNet()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_segnet_state_dict(seed).items()}
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == synth.segnet_spec()
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval(), sd


def test_segnet_matches_reference_golden():
    g = np.load(os.path.join(G, "segnet_small.npz"))
    net, _ = _net(int(g["meta"][0]))
    y = net(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert y.shape == g["logits"].shape
    assert np.abs(y - g["logits"]).max() <= 2e-4 * np.abs(g["logits"]).max()
    assert np.array_equal(y.argmax(1), g["logits"].argmax(1)) or (y.argmax(1) != g["logits"].argmax(1)).mean() < 1e-3


def test_segnet_full_frame_vs_restatement_and_api():
    net, sd = _net(31)
    rng = np.random.Generator(np.random.PCG64(9))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[None, :, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[None, :, None, None]
    x = torch.from_numpy(((rng.integers(0, 256, (1, 3, 480, 640)).astype(np.float32) / 255.0 - mean) / std).astype(np.float32))
    y = net(x.cuda()).cpu()
    with torch.no_grad():
        want = segnet_ref.segnet_forward(sd, x)
    assert y.shape == (1, 22, 480, 640)
    scale = float(want.abs().max())
    err = (y - want).abs()
    # The 2x2 max-pool positions are arg-max decisions: a 1e-7 difference flips a near-tie, the un-pooling then puts that value
    # one pixel away and the decoder convs spread it.  With synthetic weights on a noise image ~3 % of the logits move by more
    # than 5e-4 of the scale and 0.09 % of the labels change -- between the fp32 and an fp64 run of the SAME CPU restatement
    # (measured: 3.06 % / 0.76 % above 5e-4 / 5e-3, label mismatch 9.0e-4).  The HIP path has to stay inside that envelope.
    assert float(err.median()) < 2e-6 * scale
    assert float((err > 5e-4 * scale).float().mean()) < 0.05 and float((err > 5e-3 * scale).float().mean()) < 0.015
    assert float((y.argmax(1) != want.argmax(1)).float().mean()) < 2e-3
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 100, 64).cuda())                                    # H not a multiple of 32
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))                                            # CPU tensor
    net.train()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 64, 64).cuda())


def test_pool_unpool_kernels_match_torch():
    from densefusion_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randint(0, 4, (2, 8, 12, 16), device=dev).float()                    # many ties: the first maximum must win
    y = torch.empty(2, 4, 6, 16, device=dev); idx = torch.empty(2, 4, 6, 16, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    _lib.check(L.df_maxpool2x2_idx(x.data_ptr(), y.data_ptr(), idx.data_ptr(), 2, 8, 12, 16, _lib.current_stream()), "pool")
    py, pidx = F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2, return_indices=True)
    assert torch.equal(y.permute(0, 3, 1, 2), py)
    up = torch.empty(2, 8, 12, 16, device=dev)
    _lib.check(L.df_maxunpool2x2(y.data_ptr(), idx.data_ptr(), up.data_ptr(), 2, 4, 6, 16, _lib.current_stream()), "unpool")
    assert torch.equal(up.permute(0, 3, 1, 2), F.max_unpool2d(py, pidx, 2, 2))
