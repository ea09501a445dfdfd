"""Oracle: CPU restatement of the ADD / ADD-S losses (torch fp32).  TEST INFRASTRUCTURE.

Follows lib/loss.py:13-70 (PoseNet loss) and lib/loss_refiner.py:12-62 (refiner loss).  In this
fork both files import ``nn_distance`` from lib/nn.py but still call it with the CUDA-KNN
convention (lib/loss.py:44-45), which raises for symmetric objects; the *intended* semantics are
those of lib/knn (commented import lib/loss.py:9, live use tools/eval_linemod.py:123-128):
``inds = knn(target[1,3,M], pred[1,3,Q])`` -> 1-based nearest target index per pred point.
That is what the symmetric branch below implements, through oracle/knn_ref.c.
"""
from __future__ import annotations

import torch

from .knn import knn_ref


def quat_to_rot_rows(q: torch.Tensor) -> torch.Tensor:
    """lib/loss.py:18-26: the 9 entries, row-major, of R(q) for unit quats q [P,4] -> [P,3,3]."""
    a, b, c, d = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = torch.stack((1.0 - 2.0 * (c ** 2 + d ** 2), 2.0 * b * c - 2.0 * a * d, 2.0 * a * c + 2.0 * b * d,
                        2.0 * b * c + 2.0 * d * a, 1.0 - 2.0 * (b ** 2 + d ** 2), -2.0 * a * b + 2.0 * c * d,
                        -2.0 * a * c + 2.0 * b * d, 2.0 * a * b + 2.0 * c * d, 1.0 - 2.0 * (b ** 2 + c ** 2)),
                       dim=1)
    return rows.view(-1, 3, 3)


def _nearest_target(target_m3: torch.Tensor, pred_pm3: torch.Tensor) -> torch.Tensor:
    """Replace every pred point's target by its nearest target point (lib/loss.py:41-47)."""
    P, M, _ = pred_pm3.shape
    ref = target_m3.t().contiguous().numpy()[None]                      # [1,3,M]
    qry = pred_pm3.detach().permute(2, 0, 1).contiguous().view(3, -1).numpy()[None]   # [1,3,P*M]; the match is a constant
    inds = torch.from_numpy(knn_ref(ref, qry, 1)[0, 0] - 1)
    return target_m3[inds].view(P, M, 3)


def loss_calculation(pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine,
                     num_point_mesh, sym_list):
    """lib/loss.py:13-70.  Shapes: pred_r [1,N,4], pred_t [1,N,3], pred_c [1,N,1],
    target/model_points [1,M,3], idx [1,1], points [1,N,3]."""
    bs, num_p, _ = pred_c.shape
    q = (pred_r / torch.norm(pred_r, dim=2).view(bs, num_p, 1)).view(bs * num_p, 4)
    ori_base = quat_to_rot_rows(q)
    base = ori_base.transpose(2, 1).contiguous()
    mp = model_points.view(1, num_point_mesh, 3).expand(num_p, num_point_mesh, 3)
    tg0 = target.view(num_point_mesh, 3)
    tgt = tg0.view(1, num_point_mesh, 3).expand(num_p, num_point_mesh, 3)
    pt = pred_t.contiguous().view(bs * num_p, 1, 3)
    pts = points.contiguous().view(bs * num_p, 1, 3)
    c = pred_c.contiguous().view(bs * num_p)
    pred = torch.bmm(mp, base) + (pts + pt)
    if not refine and int(idx.reshape(-1)[0]) in sym_list:
        tgt = _nearest_target(tg0, pred)
    dis = torch.mean(torch.norm(pred - tgt, dim=2), dim=1)
    loss = torch.mean(dis * c - w * torch.log(c), dim=0)
    which = int(torch.max(c.view(bs, num_p), 1)[1][0])
    t = pt[which] + pts[which]                               # [1,3]
    Rsel = ori_base[which].view(1, 3, 3).contiguous()
    new_points = torch.bmm(pts.view(1, num_p, 3) - t.view(1, 1, 3), Rsel).contiguous()
    new_target = torch.bmm(tg0.view(1, num_point_mesh, 3) - t.view(1, 1, 3), Rsel).contiguous()
    return loss, dis[which], new_points.detach(), new_target.detach()      # lib/loss.py:70


def loss_refine_calculation(pred_r, pred_t, target, model_points, idx, points, num_point_mesh, sym_list):
    """lib/loss_refiner.py:12-62.  pred_r [1,4], pred_t [1,3], points [1,N,3]."""
    n_in = points.shape[1]
    q = pred_r.view(1, 4)
    q = q / torch.norm(q, dim=1).view(1, 1)
    ori_base = quat_to_rot_rows(q)
    base = ori_base.transpose(2, 1).contiguous()
    tg0 = target.view(num_point_mesh, 3)
    tgt = tg0.view(1, num_point_mesh, 3)
    t = pred_t.view(1, 1, 3)
    pred = torch.bmm(model_points.view(1, num_point_mesh, 3), base) + t
    if int(idx.reshape(-1)[0]) in sym_list:
        tgt = _nearest_target(tg0, pred)
    dis = torch.mean(torch.norm(pred - tgt, dim=2), dim=1)
    new_points = torch.bmm(points.view(1, n_in, 3) - t, ori_base).contiguous()
    new_target = torch.bmm(tg0.view(1, num_point_mesh, 3) - t, ori_base).contiguous()
    return dis, new_points.detach(), new_target.detach()      # lib/loss_refiner.py:62
