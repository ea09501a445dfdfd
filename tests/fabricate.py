"""Fabricated dataset trees shared by the dataset tests and by oracle/make_golden_datasets.py (which runs the REFERENCE's loaders over
the very same trees in the build container): YCB-Video- and LineMOD-shaped directories of random 480x640 frames written from seeded
numpy generators -- PNGs are lossless, so the GPU box rebuilds bit-identical trees from the same seeds."""
import os

import numpy as np
import scipy.io as scio
import yaml
from PIL import Image

from densefusion_amd import synth

CLASSES = ["002_master_chef_can", "003_cracker_box", "004_sugar_box", "005_tomato_soup_can"]


def make_ycb_tree(root, cfg, rng):
    os.makedirs(cfg)
    with open(f"{cfg}/classes.txt", "w") as f:
        f.write("\n".join(CLASSES) + "\n")
    for c in CLASSES:
        os.makedirs(f"{root}/models/{c}")
        np.savetxt(f"{root}/models/{c}/points.xyz", (rng.random((2700, 3)) - 0.5) * 0.2, fmt="%.6f")
    names = []
    for seq, frames in (("0001", 3), ("0060", 3)):
        os.makedirs(f"{root}/data/{seq}")
        for fr in range(frames):
            name = f"data/{seq}/{fr + 1:06d}"
            names.append(name)
            rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            depth = rng.integers(4000, 15000, (480, 640)).astype(np.uint16)
            depth[rng.random((480, 640)) < 0.08] = 0
            label = np.zeros((480, 640), dtype=np.uint8)
            present = [1, 3, 4] if fr % 2 == 0 else [2, 4]
            for k, c in enumerate(present):
                h, w = int(rng.integers(40, 200)), int(rng.integers(40, 260))
                r0, c0 = int(rng.integers(0, 480 - h)), int(rng.integers(0, 640 - w))
                label[r0:r0 + h, c0:c0 + w][rng.random((h, w)) < 0.7] = c
            if fr == 1:
                label[label == 4] = 0
                label[5:9, 5:12] = 4                                  # object 4: 28 pixels only -> never selected (minimum 50)
            poses = np.stack([np.concatenate([synth.quat_to_rot(synth.random_unit_quaternion(rng)), rng.normal(size=(3, 1)) * 0.2 + [[0], [0], [1.0]]], axis=1)
                              for _ in present], axis=2)
            Image.fromarray(rgb).save(f"{root}/{name}-color.png")
            Image.fromarray(depth).save(f"{root}/{name}-depth.png")
            Image.fromarray(label).save(f"{root}/{name}-label.png")
            scio.savemat(f"{root}/{name}-meta.mat", {"cls_indexes": np.array(present, dtype=np.uint8)[:, None], "poses": poses,
                                                     "factor_depth": np.array([[10000]], dtype=np.uint16)})
    os.makedirs(f"{root}/data_syn")
    syn = []
    for fr in range(2):                                           # synthetic frames: RGBA renders on black, three objects each
        name = f"data_syn/{fr:06d}"
        syn.append(name)
        label = np.zeros((480, 640), dtype=np.uint8)
        present = [1, 2, 3]
        for c in present:
            h, w = int(rng.integers(150, 260)), int(rng.integers(200, 330))
            r0, c0 = int(rng.integers(0, 480 - h)), int(rng.integers(0, 640 - w))
            label[r0:r0 + h, c0:c0 + w][rng.random((h, w)) < 0.8] = c
        rgba = rng.integers(0, 256, (480, 640, 4), dtype=np.uint8)
        rgba[label == 0] = 0
        rgba[..., 3] = np.where(label > 0, 255, 0)
        depth = rng.integers(4000, 15000, (480, 640)).astype(np.uint16)
        poses = np.stack([np.concatenate([synth.quat_to_rot(synth.random_unit_quaternion(rng)), rng.normal(size=(3, 1)) * 0.2 + [[0], [0], [1.0]]], axis=1)
                          for _ in present], axis=2)
        Image.fromarray(rgba).save(f"{root}/{name}-color.png")
        Image.fromarray(depth).save(f"{root}/{name}-depth.png")
        Image.fromarray(label).save(f"{root}/{name}-label.png")
        scio.savemat(f"{root}/{name}-meta.mat", {"cls_indexes": np.array(present, dtype=np.uint8)[:, None], "poses": poses,
                                                 "factor_depth": np.array([[10000]], dtype=np.uint16)})
    with open(f"{cfg}/test_data_list.txt", "w") as f:
        f.write("\n".join(names) + "\n")
    with open(f"{cfg}/train_data_list.txt", "w") as f:
        f.write("\n".join(names[:4] + syn) + "\n")
    return names


OBJLIST = [1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15]


def _write_ply(path, pts):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment fabricated\nelement vertex %d\n" % len(pts))
        f.write("property float x\nproperty float y\nproperty float z\nproperty uchar red\nend_header\n")
        for p in pts:
            f.write("%.6f %.6f %.6f 255\n" % (p[0], p[1], p[2]))


def make_linemod_tree(root, frames_per_obj=12, seed=0):
    """A LineMOD-shaped tree: 13 objects, a few 480x640 frames each (blob mask + a distractor blob, depth with holes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    os.makedirs(f"{root}/models")
    for obj in OBJLIST:
        sub = "%02d" % obj
        for d in ("rgb", "depth", "mask"):
            os.makedirs(f"{root}/data/{sub}/{d}")
        os.makedirs(f"{root}/segnet_results/{sub}_label")
        _write_ply(f"{root}/models/obj_{sub}.ply", rng.uniform(-60, 60, size=(640, 3)))
        names, gt = [], {}
        for k in range(frames_per_obj):
            name = "%04d" % (k * 3)
            names.append(name)
            rgb = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
            depth = rng.integers(400, 1500, size=(480, 640)).astype(np.uint16)
            depth[rng.random((480, 640)) < 0.1] = 0
            mask = np.zeros((480, 640), dtype=np.uint8)
            bh, bw = int(rng.integers(30, 170)), int(rng.integers(30, 220))
            r0, c0 = int(rng.integers(0, 480 - bh)), int(rng.integers(0, 640 - bw))
            if k == 1:
                r0, c0 = 0, 640 - bw                      # box touching two frame edges
            blob = rng.random((bh, bw)) < (0.9 if k != 2 else 0.02)       # frame 2: fewer mask pixels than num_points -> wrap padding
            blob[0, :] = blob[-1, :] = True
            blob[:, 0] = blob[:, -1] = True
            mask[r0:r0 + bh, c0:c0 + bw][blob] = 255
            lab = mask.copy()
            lab[5:12, 5:11] = 255                         # a small false-positive blob in the segmentation result
            if k == 3:
                lab[:] = 0                                # segmentation lost the object
            Image.fromarray(rgb).save(f"{root}/data/{sub}/rgb/{name}.png")
            Image.fromarray(depth).save(f"{root}/data/{sub}/depth/{name}.png")
            Image.fromarray(np.stack([mask] * 3, axis=2)).save(f"{root}/data/{sub}/mask/{name}.png")
            Image.fromarray(lab).save(f"{root}/segnet_results/{sub}_label/{name}_label.png")
            R = synth.quat_to_rot(synth.random_unit_quaternion(rng))
            entry = {"cam_R_m2c": [float(v) for v in R.reshape(-1)], "cam_t_m2c": [float(v) for v in rng.uniform(-100, 100, 3) + [0, 0, 900]],
                     "obj_bb": [c0, r0, bw, bh], "obj_id": obj}
            gt[k * 3] = [{"cam_R_m2c": [0.0] * 9, "cam_t_m2c": [0.0] * 3, "obj_bb": [1, 1, 50, 50], "obj_id": 9}, entry] if obj == 2 else [entry]
        for lst in ("train", "test"):
            with open(f"{root}/data/{sub}/{lst}.txt", "w") as f:
                f.write("\n".join(names) + "\n")
        with open(f"{root}/data/{sub}/gt.yml", "w") as f:
            yaml.safe_dump(gt, f)
    return root
