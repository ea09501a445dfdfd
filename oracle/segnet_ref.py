"""Oracle: torch-CPU functional restatement of the reference SegNet forward in eval mode.  TEST INFRASTRUCTURE.

Follows vanilla_segmentation/segnet.py:73-121: 13 x (conv3x3 pad 1 -> BatchNorm (running statistics) -> ReLU) with a
2x2 / stride-2 max-pool (indices kept) after blocks of 2, 2, 3, 3, 3 layers, then the mirror image: max-unpool with the
matching indices, convs back down, the last conv without BatchNorm / ReLU.  Pinned by a golden produced from the imported
reference module (tests/golden/segnet_small.npz, oracle/make_golden.py).
"""
from __future__ import annotations

import torch.nn.functional as F

_ENC_BLOCKS = [["11", "12"], ["21", "22"], ["31", "32", "33"], ["41", "42", "43"], ["51", "52", "53"]]
_DEC_BLOCKS = [["53d", "52d", "51d"], ["43d", "42d", "41d"], ["33d", "32d", "31d"], ["22d", "21d"], ["12d", "11d"]]


def _cbr(sd, name, x, relu_bn=True):
    x = F.conv2d(x, sd[f"conv{name}.weight"], sd[f"conv{name}.bias"], padding=1)
    if relu_bn:
        x = F.batch_norm(x, sd[f"bn{name}.running_mean"], sd[f"bn{name}.running_var"], sd[f"bn{name}.weight"], sd[f"bn{name}.bias"],
                         training=False, momentum=0.1, eps=1e-5)
        x = F.relu(x)
    return x


def segnet_forward(sd, x):
    """sd: {key: torch tensor} in the reference layout; x [B,3,H,W] -> logits [B,label_nbr,H,W]."""
    ids = []
    for block in _ENC_BLOCKS:                                  # segnet.py:75-98
        for name in block:
            x = _cbr(sd, name, x)
        x, idx = F.max_pool2d(x, kernel_size=2, stride=2, return_indices=True)
        ids.append(idx)
    for block in _DEC_BLOCKS:                                  # segnet.py:100-119
        x = F.max_unpool2d(x, ids.pop(), kernel_size=2, stride=2)
        for name in block:
            x = _cbr(sd, name, x, relu_bn=(name != "11d"))
    return x
