"""``PoseDataset`` for LineMOD -- host-side mirror of datasets/linemod/dataset.py:24-209 over the device-side input
preparation (``df_preprocess_objects``).

Kept from the reference: constructor ``PoseDataset(mode, num, add_noise, root, noise_trans, refine)``, the directory
layout it reads (``data/XX/{train,test}.txt``, ``rgb|depth|mask/NNNN.png``, ``segnet_results/XX_label/NNNN_label.png``
in 'eval' mode, ``data/XX/gt.yml``, ``models/obj_XX.ply``), the every-10th-line rule of 'test' mode (:48-50), the
6-tuple of ``__getitem__`` (cloud [N,3], choose [1,N] int64, img [3,H,W], target [M,3], model_points [M,3], idx [1])
with the six-``LongTensor([0])`` sentinel when no mask pixel is in the crop (:135-137), ``get_sym_list()`` and
``get_num_points_mesh()``, the camera constants and the box snapping (``get_bbox`` :235-277, integer arithmetic).

Different by design: the host only decodes the PNGs and finds the box; mask, ``choose`` sampling, back-projection
and the normalised crop of ALL requested frames of one crop size run as one HIP launch (``batch()``), and the tensors
come back resident on the device (``.cuda()`` on them is a no-op).  The random pixel subset follows the key rule of
include/dfusion.h; ``mask_to_bbox`` uses 8-connected components instead of OpenCV contours (same rectangles:
findContours traces the outer border of exactly those components).  ``add_noise=True`` (:114-115,159-160,178-180): colour
jitter on the decoded frame (``augment.ColorJitter``, a restatement of the pinned torchvision's) and one random translation
added to cloud and target.
"""
from __future__ import annotations

import random

import numpy as np
import torch
import yaml
from PIL import Image
from scipy import ndimage

from ...lib import preprocess as pp
from .. import augment

OBJLIST = [1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15]
IMG_H, IMG_W = 480, 640


def ply_vtx(path):
    """Vertex coordinates of an ASCII .ply (datasets/linemod/dataset.py:280-291): element count on header line 4,
    x y z = the first three columns of each vertex row.  float32 [n,3]."""
    with open(path) as f:
        head = [f.readline() for _ in range(4)]
        if head[0].strip() != "ply":
            raise ValueError(f"{path}: not a ply file")
        n = int(head[3].split()[-1])
        for line in f:
            if line.strip() == "end_header":
                break
        rows = [f.readline().split()[:3] for _ in range(n)]
    return np.asarray(rows, dtype=np.float32).reshape(n, 3)


def mask_to_bbox(mask):
    """[x, y, w, h] of the largest bounding rectangle among the mask's blobs (dataset.py:216-232)."""
    lab, _ = ndimage.label(np.asarray(mask, dtype=np.uint8), structure=np.ones((3, 3), dtype=int))
    best, area = (0, 0, 0, 0), 0
    for rows, cols in ndimage.find_objects(lab):
        w, h = cols.stop - cols.start, rows.stop - rows.start
        if w * h > area:
            best, area = (cols.start, rows.start, w, h), w * h
    return list(best)


def get_bbox(bbox):
    """[x, y, w, h] -> (rmin, rmax, cmin, cmax) snapped to the border list and kept inside 480x640 (dataset.py:235-277)."""
    rmin, rmax = max(bbox[1], 0), min(bbox[1] + bbox[3], IMG_H - 1)
    cmin, cmax = max(bbox[0], 0), min(bbox[0] + bbox[2], IMG_W - 1)
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


def _load_yaml(path):
    loader = getattr(yaml, "CSafeLoader", yaml.SafeLoader)
    with open(path, "r") as f:
        return yaml.load(f, Loader=loader)


class PoseDataset:
    def __init__(self, mode, num, add_noise, root, noise_trans, refine, device="cuda", seed=0):
        if mode not in ("train", "test", "eval"):
            raise ValueError(f"mode must be train / test / eval, got {mode!r}")
        self.objlist = list(OBJLIST)
        self.mode, self.num, self.root, self.refine = mode, int(num), root, refine
        self.noise_trans, self.add_noise = noise_trans, bool(add_noise)
        self.trancolor = augment.ColorJitter(0.2, 0.2, 0.2, 0.05)                  # :83
        self.device = torch.device(device)
        self.seed = int(seed)
        self.list_rgb, self.list_depth, self.list_label, self.list_obj, self.list_rank = [], [], [], [], []
        self.meta, self.pt = {}, {}
        for item in self.objlist:
            sub = "%02d" % item
            with open(f"{root}/data/{sub}/{'train' if mode == 'train' else 'test'}.txt") as f:
                names = [ln.rstrip("\n") for ln in f if ln.strip()]
            if mode == "test":
                names = names[9::10]            # the running counter of :44-52 restarts at a multiple of 10 for every object
            for name in names:
                self.list_rgb.append(f"{root}/data/{sub}/rgb/{name}.png")
                self.list_depth.append(f"{root}/data/{sub}/depth/{name}.png")
                if mode == "eval":
                    self.list_label.append(f"{root}/segnet_results/{sub}_label/{name}_label.png")
                else:
                    self.list_label.append(f"{root}/data/{sub}/mask/{name}.png")
                self.list_obj.append(item)
                self.list_rank.append(int(name))
            self.meta[item] = _load_yaml(f"{root}/data/{sub}/gt.yml")
            self.pt[item] = ply_vtx(f"{root}/models/obj_{sub}.ply")
        self.length = len(self.list_rgb)
        self.num_pt_mesh_large = self.num_pt_mesh_small = 500
        self.symmetry_obj_idx = [7, 8]

    def __len__(self):
        return self.length

    def get_sym_list(self):
        return self.symmetry_obj_idx

    def get_num_points_mesh(self):
        return self.num_pt_mesh_large if self.refine else self.num_pt_mesh_small

    # -- host part: decode, find the box -------------------------------------------------------------------------
    def _meta(self, obj, rank):
        entries = self.meta[obj][rank]
        if obj == 2:                              # the only sequence whose gt.yml lists several objects per frame (:98-104)
            for e in entries:
                if e["obj_id"] == 2:
                    return e
        return entries[0]

    def _host_frame(self, index):
        img = Image.open(self.list_rgb[index])
        if self.add_noise:
            img = self.trancolor(img)                         # :114-115
        rgb = np.asarray(img)[:, :, :3]
        depth = np.asarray(Image.open(self.list_depth[index])).astype(np.uint16)
        label = np.asarray(Image.open(self.list_label[index]))
        obj, rank = self.list_obj[index], self.list_rank[index]
        meta = self._meta(obj, rank)
        if self.mode == "eval":
            lab2d = label if label.ndim == 2 else label[:, :, 0]
            box = get_bbox(mask_to_bbox(lab2d == 255))
        else:
            lab2d = label[:, :, 0] if label.ndim == 3 else label       # :110 keeps channel 0 of the per-channel comparison
            box = get_bbox(meta["obj_bb"])
        return np.ascontiguousarray(rgb), depth, np.ascontiguousarray(lab2d).astype(np.int32), box, obj, meta

    def _targets(self, obj, meta, add_t=None):
        pts = self.pt[obj] / 1000.0
        n, keep_n = len(pts), self.num_pt_mesh_small
        keep = np.ones(n, dtype=bool)
        keep[random.sample(range(n), n - keep_n)] = False                  # same draw as :167-170 on Python's global stream
        model_points = pts[keep]                                           # (np.delete keeps the survivors in index order: so does this)
        target_r = np.resize(np.array(meta["cam_R_m2c"]), (3, 3))
        target = np.dot(model_points, target_r.T) + np.array(meta["cam_t_m2c"]) / 1000.0
        if add_t is not None:
            target = target + add_t                           # :178-179
        return torch.from_numpy(target.astype(np.float32)), torch.from_numpy(model_points.astype(np.float32))

    # -- device part: one launch per crop size -----------------------------------------------------------------
    def batch(self, indices):
        """The 6-tuples of ``indices`` (same order), every crop size prepared by one device launch."""
        host = [self._host_frame(i) for i in indices]
        dev = self.device
        rgb = torch.from_numpy(np.stack([h[0] for h in host])).to(dev)
        depth = torch.from_numpy(np.stack([h[1] for h in host]).view(np.int16)).to(dev)
        label = torch.from_numpy(np.stack([h[2] for h in host])).to(dev)
        groups = {}
        for k, h in enumerate(host):
            rmin, rmax, cmin, cmax = h[3]
            groups.setdefault((rmax - rmin, cmax - cmin), []).append(k)
        prepared = [None] * len(indices)
        for members in groups.values():
            objs = [(k, 255, host[k][3], (self.seed * 1000003 + indices[k]) & 0xFFFFFFFF) for k in members]
            img, cloud, choose, count = pp.preprocess_objects(rgb, depth, label, objs, self.num, cam=pp.LINEMOD_CAM)
            for j, (k, c) in enumerate(zip(members, count.tolist())):
                prepared[k] = (cloud[j], choose[j], img[j]) if c else None
        out = []
        for k, h in enumerate(host):                     # index order: Python's global random stream is consumed like :129,167-170
            add_t = np.array([random.uniform(-self.noise_trans, self.noise_trans) for _ in range(3)])      # drawn even when unused (:132)
            if prepared[k] is None:
                cc = torch.LongTensor([0])
                out.append((cc, cc, cc, cc, cc, cc))
                continue
            target, model_points = self._targets(h[4], h[5], add_t if self.add_noise else None)
            idx = torch.tensor([self.objlist.index(h[4])], dtype=torch.int64, device=dev)
            idx._host = [int(self.objlist.index(h[4]))]          # the trainer's losses branch on the index: spare it a device read-back
            cloud, choose, img = prepared[k]
            if self.add_noise:
                cloud = cloud + torch.from_numpy(add_t.astype(np.float32)).to(dev)      # :159-160
            out.append((cloud, choose, img, target.to(dev), model_points.to(dev), idx))
        return out

    def __getitem__(self, index):
        return self.batch([index])[0]

    # -- the same fetch cut in two for train_utils.Prefetcher(processes=...): worker processes decode, the trainer's process uploads --
    def host_item(self, index):
        """CPU half of ``__getitem__`` (never touches the device: runs in the loader's worker processes like the reference's
        DataLoader workers, tools/train.py:106): the decoded frame, the snapped box, the number of mask pixels in it and the
        sampled model / target points, as host tensors."""
        rgb, depth, lab2d, box, obj, meta = self._host_frame(index)
        rmin, rmax, cmin, cmax = box
        count = int(np.count_nonzero((depth[rmin:rmax, cmin:cmax] != 0) & (lab2d[rmin:rmax, cmin:cmax] == 255)))
        add_t = np.array([random.uniform(-self.noise_trans, self.noise_trans) for _ in range(3)])      # drawn even when unused (:132)
        target, model_points = self._targets(obj, meta, add_t if self.add_noise else None)
        return (torch.from_numpy(np.array(rgb)), torch.from_numpy(depth.view(np.int16)), torch.from_numpy(lab2d),
                torch.tensor([rmin, rmax, cmin, cmax, count, self.objlist.index(obj)], dtype=torch.int64), target, model_points,
                torch.from_numpy((add_t if self.add_noise else np.zeros(3)).astype(np.float32)))

    def device_item(self, index, host, choose=None):
        """Device half: uploads + one preparation launch on the current stream, no read-back.  Same 6-tuple as ``__getitem__``.
        ``choose``: the pixel subset as an input (tests: the reference's own draw) instead of the device-side sampling."""
        rgb, depth, lab2d, info, target, model_points, add_t = host
        rmin, rmax, cmin, cmax, count, oi = (int(v) for v in info.tolist())
        if count == 0:
            cc = torch.LongTensor([0])
            return (cc, cc, cc, cc, cc, cc)
        dev = self.device
        up = lambda t: t.to(dev, non_blocking=True)            # asynchronous when the loader pinned `t`, staged otherwise
        img, cloud, choose, _ = pp.preprocess_objects(up(rgb)[None], up(depth)[None], up(lab2d)[None],
                                                      [(0, 255, (rmin, rmax, cmin, cmax), (self.seed * 1000003 + int(index)) & 0xFFFFFFFF)],
                                                      self.num, cam=pp.LINEMOD_CAM, choose_in=choose)
        idx = torch.tensor([oi], dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        idx._host = [oi]
        if self.add_noise:
            cloud = cloud + up(add_t)                         # :159-160 (the same translation went into the target)
        return (cloud[0], choose[0], img[0], up(target), up(model_points), idx)
