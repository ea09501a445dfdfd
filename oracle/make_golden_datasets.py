"""Generate tests/golden/datasets_*.npz by running the REFERENCE's own loaders and box helpers on CPU.

Dev-only script for the build container, where /root/reference exists; the GPU box only sees the committed fixtures and rebuilds the
same fabricated dataset trees from their seeds (tests/fabricate.py: PNG is lossless).  Nothing from the reference is copied: its
modules are loaded from where they lie, run, and their OUTPUTS are stored.

What is imported and how:
  * ``datasets/ycb/dataset.py`` and ``datasets/linemod/dataset.py`` are loaded by file path (the name ``datasets`` belongs to an
    installed package here), with /root/reference on sys.path for their ``lib.transformations`` import.
  * Absent libraries get placeholders so that the imports proceed: ``torchvision.transforms`` (``ColorJitter`` is constructed but
    never called with ``add_noise=False`` -- the placeholder raises if it is; ``Normalize`` IS called on the crop: the placeholder
    applies torchvision's documented arithmetic ``(x - mean[c]) / std[c]``, so the golden ``img`` pins everything in the loaders but
    that one library line), ``cv2`` (only ``mask_to_bbox`` of the LineMOD 'eval' mode uses it: not run, stays unpinned).
  * ``yaml.load(f)`` without a Loader (datasets/linemod/dataset.py:66) raises under PyYAML 6: while the LineMOD module is imported the
    name ``yaml`` resolves to a shim whose ``load`` is ``yaml.safe_load`` (the fabricated gt.yml holds plain lists and numbers).
  * ``tools/eval_ycb.py`` runs a whole evaluation at import; only its ``get_bbox`` (:54-90) and the three constants it reads are
    taken out of the parsed source (``ast``) and executed.

RNG: before every ``__getitem__`` the script seeds ``np.random`` and ``random`` (the two global streams the loaders draw from); the
tests seed them the same way.  The product draws the object slot, the translation noise and the model-point subset from the same
streams in the same order, so those match the reference by themselves; the pixel subset of ``choose`` comes from
``np.random.shuffle`` (datasets/ycb/dataset.py:174-178), which a GPU cannot share: the tests hand the golden ``choose`` to the device
preparation as an input (include/dfusion.h, df_preprocess_objects ``given``) and compare everything computed from it.

    python -m oracle.make_golden_datasets            # writes tests/golden/datasets_{ycb,linemod,bbox}.npz
"""
from __future__ import annotations

import ast
import importlib.util
import os
import random
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("DF_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

YCB_TREE_SEED, LINEMOD_TREE_SEED = 3, 0          # the seeds the tests rebuild the trees from
YCB_N, LINEMOD_N = 1000, 500


def seed_for(kind, i):
    """(np.random seed, random seed) in front of item i -- shared with the tests."""
    return {"ycb": (100 + i, 200 + i), "linemod": (300 + i, 400 + i)}[kind]


def _placeholders():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class ColorJitter:                                  # constructed by the loaders, called only with add_noise=True
        def __init__(self, *a, **k):
            pass

        def __call__(self, img):
            raise RuntimeError("ColorJitter placeholder called: the goldens are made with add_noise=False")

    class Normalize:                                    # torchvision.transforms.Normalize: (x - mean[c]) / std[c] on a [C,H,W] tensor
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    tr.ColorJitter, tr.Normalize = ColorJitter, Normalize
    tv.transforms = tr
    for name, mod in (("torchvision", tv), ("torchvision.transforms", tr), ("torchvision.utils", types.ModuleType("torchvision.utils")),
                      ("torchvision.datasets", types.ModuleType("torchvision.datasets")), ("cv2", types.ModuleType("cv2"))):
        sys.modules.setdefault(name, mod)


def _load(path, name, yaml_shim=False):
    real_yaml = sys.modules.get("yaml")
    if yaml_shim:
        import yaml as _y
        shim = types.ModuleType("yaml")
        shim.__dict__.update({k: getattr(_y, k) for k in dir(_y) if not k.startswith("__")})
        shim.load = lambda stream, Loader=None: _y.safe_load(stream)
        sys.modules["yaml"] = shim
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if yaml_shim and real_yaml is not None:
            sys.modules["yaml"] = real_yaml
    return mod


def _img_summary(img):
    """The crop is up to 480x640x3 floats: keep its shape, a 1-in-9 sub-sample (every third row and column: shifts by one pixel change
    it), per-channel float64 sums and the sum of squares."""
    a = img.numpy().astype(np.float32)
    return dict(img_shape=np.array(a.shape, dtype=np.int32), img_sub=a[:, ::3, ::3].copy(), img_sum=a.astype(np.float64).sum(axis=(1, 2)),
                img_sq=np.array(float((a.astype(np.float64) ** 2).sum())))


def ycb_goldens(work):
    import fabricate
    root, cfg_src = os.path.join(work, "YCB"), os.path.join(work, "cfg")
    names = fabricate.make_ycb_tree(root, cfg_src, np.random.default_rng(YCB_TREE_SEED))
    # the reference opens 'datasets/ycb/dataset_config/...' relative to the working directory (datasets/ycb/dataset.py:21-23,50)
    cwd = os.path.join(work, "cwd")
    os.makedirs(os.path.join(cwd, "datasets", "ycb"))
    shutil.copytree(cfg_src, os.path.join(cwd, "datasets", "ycb", "dataset_config"))
    mod = _load(os.path.join(REF, "datasets", "ycb", "dataset.py"), "ref_ycb_dataset")
    out = {"names": np.array(names)}
    old = os.getcwd()
    os.chdir(cwd)
    try:
        for refine in (False, True):
            ds = mod.PoseDataset("test", YCB_N, False, root, 0.0, refine)
            assert len(ds) == len(names)
            tag = "r1" if refine else "r0"
            for i in (range(len(names)) if not refine else (0, 3)):
                ns, rs = seed_for("ycb", i)
                np.random.seed(ns); random.seed(rs)
                cloud, choose, img, target, model_points, idx = ds[i]
                out.update({f"{tag}_{i}_cloud": cloud.numpy(), f"{tag}_{i}_choose": choose.numpy().astype(np.int32), f"{tag}_{i}_target": target.numpy(),
                            f"{tag}_{i}_model_points": model_points.numpy(), f"{tag}_{i}_idx": idx.numpy().astype(np.int32)})
                out.update({f"{tag}_{i}_{k}": v for k, v in _img_summary(img).items()})
    finally:
        os.chdir(old)
    # get_bbox(label) on a sweep of masks (datasets/ycb/dataset.py:251-289): rectangles with ragged fill, regenerated from the seed
    rng = np.random.default_rng(41)
    rects, boxes = [], []
    for _ in range(200):
        h, w = int(rng.integers(1, 481)), int(rng.integers(1, 641))
        r0, c0 = int(rng.integers(0, 481 - h)), int(rng.integers(0, 641 - w))
        m = np.zeros((480, 640), dtype=bool)
        m[r0, c0:c0 + w] = True; m[r0:r0 + h, c0] = True; m[r0 + h - 1, c0 + w - 1] = True
        rects.append((r0, c0, h, w))
        boxes.append(mod.get_bbox(m))
    out["bbox_rects"] = np.array(rects, dtype=np.int32)
    out["bbox_out"] = np.array(boxes, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "datasets_ycb.npz"), **out)
    print("datasets_ycb.npz:", len(out), "arrays")


def linemod_goldens(work):
    import fabricate
    root = fabricate.make_linemod_tree(os.path.join(work, "LM"), frames_per_obj=12, seed=LINEMOD_TREE_SEED)
    mod = _load(os.path.join(REF, "datasets", "linemod", "dataset.py"), "ref_linemod_dataset", yaml_shim=True)
    out = {}
    for mode, picks in (("test", None), ("train", (0, 1, 2, 14))):
        ds = mod.PoseDataset(mode, LINEMOD_N, False, root, 0.0, True)
        idxs = range(len(ds)) if picks is None else picks
        out[f"{mode}_len"] = np.array(len(ds))
        out[f"{mode}_items"] = np.array(list(idxs), dtype=np.int32)
        for i in idxs:
            ns, rs = seed_for("linemod", i)
            np.random.seed(ns); random.seed(rs)
            cloud, choose, img, target, model_points, idx = ds[i]
            tag = f"{mode}_{i}"
            out.update({f"{tag}_cloud": cloud.numpy(), f"{tag}_choose": choose.numpy().astype(np.int32), f"{tag}_target": target.numpy(),
                        f"{tag}_model_points": model_points.numpy(), f"{tag}_idx": idx.numpy().astype(np.int32)})
            out.update({f"{tag}_{k}": v for k, v in _img_summary(img).items()})
    # get_bbox(bbox) sweep (datasets/linemod/dataset.py:235-277) and ply_vtx (:280-291)
    rng = np.random.default_rng(42)
    ins = np.stack([rng.integers(-30, 670, 400), rng.integers(-30, 510, 400), rng.integers(0, 660, 400), rng.integers(0, 500, 400)], axis=1).astype(np.int32)
    out["bbox_in"] = ins
    out["bbox_out"] = np.array([mod.get_bbox([int(v) for v in b]) for b in ins], dtype=np.int32)
    out["ply_obj_05"] = mod.ply_vtx(f"{root}/models/obj_05.ply")
    np.savez_compressed(os.path.join(OUT, "datasets_linemod.npz"), **out)
    print("datasets_linemod.npz:", len(out), "arrays")


def eval_ycb_bbox_golden():
    """tools/eval_ycb.py:54-90 ``get_bbox(posecnn_rois)`` (reads the globals idx, border_list, img_width, img_length)."""
    src = open(os.path.join(REF, "tools", "eval_ycb.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name == "get_bbox")
            or (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in ("border_list", "img_width", "img_length") for t in n.targets))]
    assert len(keep) == 4, [type(n).__name__ for n in keep]
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "eval_ycb_get_bbox", "exec"), ns)      # noqa: S102 - the reference's own function
    rng = np.random.default_rng(43)
    rois, outs = [], []
    for _ in range(400):
        x1, y1 = float(rng.uniform(-20, 600)), float(rng.uniform(-20, 440))
        w, h = float(rng.uniform(3, 660)), float(rng.uniform(3, 500))
        roi = [0.0, float(rng.integers(1, 22)), x1, y1, x1 + w, y1 + h, 1.0]
        ns["idx"] = 0
        rois.append(roi)
        outs.append(ns["get_bbox"]([roi]))
    np.savez_compressed(os.path.join(OUT, "datasets_bbox_eval_ycb.npz"), rois=np.array(rois), out=np.array(outs, dtype=np.int32))
    print("datasets_bbox_eval_ycb.npz: 400 rois")


def main():
    warnings.filterwarnings("ignore")
    _placeholders()
    sys.path.insert(0, REF)
    work = tempfile.mkdtemp(prefix="df_golden_")
    try:
        ycb_goldens(work)
        linemod_goldens(work)
        eval_ycb_bbox_golden()
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
