"""``PoseDataset`` for YCB-Video -- host-side mirror of datasets/ycb/dataset.py:18-247 over the device-side input preparation.

Kept from the reference: constructor ``PoseDataset(mode, num_pt, add_noise, root, noise_trans, refine)``, the list /
class files it reads (``dataset_config/{train,test}_data_list.txt``, ``classes.txt``, ``models/<class>/points.xyz``,
``<frame>-{color,depth,label}.png``, ``<frame>-meta.mat``), the random choice of one object per frame with more than
50 valid pixels (np.random.randint on numpy's global stream, :139-146), the label-extent box (``get_bbox`` :251-289), the
two camera intrinsics by sequence number (:96-105) and ``factor_depth``, the model-point subset on Python's global
``random`` stream (500 points, 2600 with ``refine``; :199-204), the 6-tuple, ``get_sym_list()`` / ``get_num_points_mesh()``,
and the training augmentation (``add_noise=True``, ``data_syn`` frames; :117-136,149-167,196-221): colour jitter (``augment.ColorJitter``,
a restatement of the pinned torchvision's), two occluding objects from a synthetic frame, a synthetic frame pasted over a random
real background, N(0, 7) pixel noise on synthetic frames, one translation added to cloud and target.

Different by design: the host composes the augmented FULL frame (the reference augments the crop: same pixels inside the box);
mask, ``choose`` sampling, back-projection and the normalised crop run on the device (``df_preprocess_objects``; the pixel-subset
rule of include/dfusion.h replaces np.random.shuffle) and the tensors stay there; the pixel noise is drawn on the device after
the (affine) normalisation; ``host_item`` / ``device_item`` cut a fetch into the half that runs in worker processes and the half
that touches the device.
"""
from __future__ import annotations

import random

import numpy as np
import scipy.io as scio
import torch
from PIL import Image

from ...lib import preprocess as pp
from .. import augment

CAM_1 = dict(cx=312.9869, cy=241.3109, fx=1066.778, fy=1067.487)          # dataset.py:71-74
CAM_2 = dict(cx=323.7872, cy=279.6921, fx=1077.836, fy=1078.189)          # :76-79, sequences >= 60
IMG_H, IMG_W = 480, 640


def get_bbox(label):
    """Boolean mask of one object -> (rmin, rmax, cmin, cmax): its extents snapped to the border list, inside 480x640
    (datasets/ycb/dataset.py:251-289)."""
    rows, cols = np.flatnonzero(np.any(label, axis=1)), np.flatnonzero(np.any(label, axis=0))
    rmin, rmax, cmin, cmax = int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1
    r_b, c_b = pp._snap(rmax - rmin), pp._snap(cmax - cmin)
    cr, cc = int((rmin + rmax) / 2), int((cmin + cmax) / 2)
    rmin, rmax = cr - int(r_b / 2), cr + int(r_b / 2)
    cmin, cmax = cc - int(c_b / 2), cc + int(c_b / 2)
    if rmin < 0:
        rmax, rmin = rmax - rmin, 0
    if cmin < 0:
        cmax, cmin = cmax - cmin, 0
    if rmax > IMG_H:
        rmin, rmax = rmin - (rmax - IMG_H), IMG_H
    if cmax > IMG_W:
        cmin, cmax = cmin - (cmax - IMG_W), IMG_W
    return rmin, rmax, cmin, cmax


class PoseDataset:
    def __init__(self, mode, num_pt, add_noise, root, noise_trans, refine, dataset_config_dir="datasets/ycb/dataset_config",
                 device="cuda", seed=0, skip_synthetic=False):
        if mode not in ("train", "test"):
            raise ValueError(f"mode must be train / test, got {mode!r}")
        self.mode, self.num_pt, self.root, self.refine = mode, int(num_pt), root, refine
        self.noise_trans, self.add_noise = noise_trans, bool(add_noise)
        self.trancolor = augment.ColorJitter(0.2, 0.2, 0.2, 0.05)                  # :84
        self.noise_img_scale, self.front_num = 7.0, 2                               # :86,93
        self.device, self.seed = torch.device(device), int(seed)
        with open(f"{dataset_config_dir}/{mode}_data_list.txt") as f:
            self.list = [ln.rstrip("\n") for ln in f if ln.strip()]
        if skip_synthetic:                                    # real frames only (the synthetic ones need the augmentation pipeline)
            self.list = [n for n in self.list if n[:5] == "data/"]
        self.real = [n for n in self.list if n[:5] == "data/"]
        self.syn = [n for n in self.list if n[:5] != "data/"]
        self.length = len(self.list)
        self.cld = {}
        with open(f"{dataset_config_dir}/classes.txt") as f:
            for class_id, name in enumerate((ln.strip() for ln in f if ln.strip()), start=1):
                self.cld[class_id] = np.loadtxt(f"{root}/models/{name}/points.xyz", dtype=np.float64).reshape(-1, 3)
        self.minimum_num_pt = 50
        self.symmetry_obj_idx = [12, 15, 18, 19, 20]
        self.num_pt_mesh_small, self.num_pt_mesh_large = 500, 2600

    def __len__(self):
        return self.length

    def get_sym_list(self):
        return self.symmetry_obj_idx

    def get_num_points_mesh(self):
        return self.num_pt_mesh_large if self.refine else self.num_pt_mesh_small

    def host_item(self, index):
        """CPU half of ``__getitem__`` (never touches the device: runs in the worker processes of
        ``train_utils.Prefetcher(processes=...)`` like the reference's DataLoader workers, tools/train.py:106): decoded frame with
        the augmentation of :117-167 composed on the FULL frame (colour jitter, synthetic frame over a real background, occluders),
        the object drawn from it, its box, camera, translation noise and the sampled model / target points, as host tensors."""
        name = self.list[index]
        syn = name[:8] == "data_syn"
        img = Image.open(f"{self.root}/{name}-color.png")
        depth = np.array(Image.open(f"{self.root}/{name}-depth.png")).astype(np.uint16)
        label = np.array(Image.open(f"{self.root}/{name}-label.png"))
        meta = scio.loadmat(f"{self.root}/{name}-meta.mat")
        cam = dict(CAM_2 if not syn and int(name[5:9]) >= 60 else CAM_1, scale=float(meta["factor_depth"][0][0]))
        mask_back = label == 0
        front, mask_front = None, None
        if self.add_noise and self.syn:                       # :117-136: two objects of a synthetic frame in front of the scene
            for _ in range(5):
                seed = random.choice(self.syn)
                cand = np.array(self.trancolor(Image.open(f"{self.root}/{seed}-color.png").convert("RGB")))
                keep = augment.occluder_mask(np.array(Image.open(f"{self.root}/{seed}-label.png")), self.front_num)
                if keep is None:
                    continue
                t_label = label * keep
                if np.count_nonzero(t_label) > 1000:
                    label, front, mask_front = t_label, cand, keep
                    break
        obj = meta["cls_indexes"].flatten().astype(np.int32)
        while True:                                           # :139-146 (an object with enough valid pixels; numpy's global stream)
            idx = np.random.randint(0, len(obj))
            mask_label = label == obj[idx]
            if np.count_nonzero(mask_label & (depth != 0)) > self.minimum_num_pt:
                break
        if self.add_noise:
            img = self.trancolor(img)                         # :149-150
        box = get_bbox(mask_label)
        rgb = np.array(img)[:, :, :3].copy()
        if syn:                                               # :155-159 (uint8 arithmetic like the reference's)
            seed = random.choice(self.real)
            back = np.array(self.trancolor(Image.open(f"{self.root}/{seed}-color.png").convert("RGB")))
            rgb = back * mask_back[:, :, None] + rgb
        if front is not None:                                 # :163-164
            rgb = rgb * mask_front[:, :, None] + front * ~mask_front[:, :, None]
        add_t = np.array([random.uniform(-self.noise_trans, self.noise_trans) for _ in range(3)])      # drawn even when unused (:171)
        pts = self.cld[int(obj[idx])]
        keep_n = self.num_pt_mesh_large if self.refine else self.num_pt_mesh_small
        keep = np.ones(len(pts), dtype=bool)
        keep[random.sample(range(len(pts)), len(pts) - keep_n)] = False               # :199-204 on Python's global stream
        model_points = pts[keep]
        pose = meta["poses"][:, :, idx]
        target = np.dot(model_points, pose[:, 0:3].T) + pose[:, 3:4].flatten()[None]
        if self.add_noise:
            target = target + add_t                           # :216-219
        return (torch.from_numpy(np.ascontiguousarray(rgb)), torch.from_numpy(depth.view(np.int16)), torch.from_numpy(label.astype(np.int32)),
                torch.tensor(list(box) + [int(obj[idx]), int(syn)], dtype=torch.int64),
                torch.tensor([cam[k] for k in ("cx", "cy", "fx", "fy", "scale")] + list(add_t if self.add_noise else np.zeros(3)), dtype=torch.float64),
                torch.from_numpy(target.astype(np.float32)), torch.from_numpy(model_points.astype(np.float32)))

    def device_item(self, index, host, choose=None):
        """Device half: uploads + one preparation launch on the current stream.  The 6-tuple of ``__getitem__``.
        ``choose``: the pixel subset as an input (tests: the reference's own draw) instead of the device-side sampling."""
        rgb, depth, label, info, camv, target, model_points = host
        rmin, rmax, cmin, cmax, cls, syn = (int(v) for v in info.tolist())
        camv = camv.tolist()
        cam, add_t = dict(zip(("cx", "cy", "fx", "fy", "scale"), camv[:5])), camv[5:8]
        dev = self.device
        up = lambda t: t.to(dev, non_blocking=True)            # asynchronous when the loader pinned `t`, staged otherwise
        img, cloud, choose, _count = pp.preprocess_objects(up(rgb)[None], up(depth)[None], up(label)[None],
                                                           [(0, cls, (rmin, rmax, cmin, cmax), (self.seed * 1000003 + int(index)) & 0xFFFFFFFF)],
                                                           self.num_pt, cam=cam, choose_in=choose)
        if any(add_t):                                       # :196-197 (the same translation went into the target)
            cloud = cloud + torch.tensor(add_t, dtype=torch.float32, device=dev)
        if syn:                                               # :166-167 N(0, 7) on the 0-255-scale pixels = N(0, 7 / std) after the normalisation
            img = img + torch.randn_like(img) * torch.tensor([self.noise_img_scale / v for v in pp.IMG_STD], device=dev).view(1, 3, 1, 1)
        index_t = torch.tensor([cls - 1], dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        index_t._host = [cls - 1]                        # the trainer's losses branch on the index: spare it a device read-back
        return (cloud[0], choose[0], img[0], up(target), up(model_points), index_t)

    def __getitem__(self, index):
        return self.device_item(index, self.host_item(index))
