"""CPU: the colour-jitter restatement (densefusion_amd/datasets/augment.py; torchvision 0.2.2.post3 is not importable here, so these
are the algorithm's own properties: draw order, neutral factors, the 8-bit hue wrap, PIL's enhancers underneath)."""
import random

import numpy as np
from PIL import Image, ImageEnhance

from densefusion_amd.datasets import augment


def _img(seed=0, mode="RGB"):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, (24, 40, 4 if mode == "RGBA" else 3), dtype=np.uint8), mode)


def test_draw_order_and_ranges(monkeypatch):
    calls = []
    real_uniform, real_shuffle = random.uniform, random.shuffle
    monkeypatch.setattr(random, "uniform", lambda a, b: calls.append(("uniform", round(a, 6), round(b, 6))) or real_uniform(a, b))
    monkeypatch.setattr(random, "shuffle", lambda x: calls.append(("shuffle", len(x))) or real_shuffle(x))
    augment.ColorJitter(0.2, 0.2, 0.2, 0.05)(_img())
    assert calls == [("uniform", 0.8, 1.2), ("uniform", 0.8, 1.2), ("uniform", 0.8, 1.2), ("uniform", -0.05, 0.05), ("shuffle", 4)]


def test_same_state_same_image_and_neutral_factors(monkeypatch):
    im = _img(1)
    random.seed(7); a = np.array(augment.ColorJitter(0.2, 0.2, 0.2, 0.05)(im))
    random.seed(7); b = np.array(augment.ColorJitter(0.2, 0.2, 0.2, 0.05)(im))
    random.seed(8); c = np.array(augment.ColorJitter(0.2, 0.2, 0.2, 0.05)(im))
    assert (a == b).all() and (a != c).any()
    assert (np.array(augment.ColorJitter()(im)) == np.array(im)).all()                    # no jitter configured: untouched, nothing drawn
    monkeypatch.setattr(random, "uniform", lambda lo, hi: (lo + hi) / 2)                  # factors 1, 1, 1, 0
    out = np.array(augment.ColorJitter(0.2, 0.2, 0.2, 0.05)(im))
    assert (out == np.array(im.convert("HSV").convert("RGB"))).all()                      # only the HSV round trip of the hue step is left


def test_single_operations_are_pils():
    im = _img(2)
    for kw, enh in (("brightness", ImageEnhance.Brightness), ("contrast", ImageEnhance.Contrast), ("saturation", ImageEnhance.Color)):
        random.seed(3)
        f = random.uniform(0.8, 1.2)
        random.seed(3)
        got = augment.ColorJitter(**{kw: 0.2})(im)
        assert (np.array(got) == np.array(enh(im).enhance(f))).all()


def test_hue_wraps_in_8_bits():
    im = _img(3)
    h0 = np.array(im.convert("HSV").split()[0]).astype(int)
    for f, shift in ((0.05, 12), (-0.05, 244), (0.5, 127), (-0.5, 129)):                  # int(f * 255) mod 256
        h1 = np.array(augment.adjust_hue(im, f).convert("HSV").split()[0]).astype(int)
        want = (h0 + shift) % 256
        # (HSV -> RGB -> HSV quantises: allow the round trip's 2 counts, modulo the wrap)
        d = np.abs((h1 - want + 128) % 256 - 128)
        sat = np.array(im.convert("HSV").split()[1])
        assert (d[sat > 40] <= 3).mean() > 0.97
    rgba = _img(4, "RGBA")
    assert augment.adjust_hue(rgba, 0.1).mode == "RGBA"
    assert augment.adjust_hue(im.convert("L"), 0.1).mode == "L"


def test_occluder_mask():
    lab = np.zeros((10, 12), dtype=np.uint8)
    lab[1:4, 1:5] = 3; lab[5:9, 2:7] = 7; lab[0:2, 8:12] = 9
    random.seed(0)
    keep = augment.occluder_mask(lab, 2)
    assert keep.dtype == bool and keep.shape == lab.shape
    gone = set(np.unique(lab[~keep]).tolist())
    assert len(gone) == 2 and gone <= {3, 7, 9} and (lab[keep] != list(gone)[0]).all()
    assert augment.occluder_mask((lab == 3).astype(np.uint8) * 3, 2) is None              # one object only: not enough to occlude with
