"""The reference's support cross-entropy as its own sequence of torch ops (test infrastructure:
the checker of the fused HIP kernel ``mz_support_ce``; only tests/ may import this).

Follows /root/reference/model.py: transform_value :84-86, scalar_to_support :112-129,
scalar_to_support_loss :132-140.  Pinned through g7_losses.npz (the reference's compute_losses).
"""
import torch
import torch.nn.functional as F


def support_ce(logits, targets, support_size: int):
    t = torch.sign(targets) * (torch.sqrt(torch.abs(targets) + 1) - 1 + 0.001 * targets)
    t = torch.clamp(t, -support_size, support_size)
    fl = t.floor()
    up_p = t - fl
    lo_p = 1 - up_p
    sup = torch.zeros(t.size(0), 2 * support_size + 1, device=t.device)
    lo = (fl + support_size).long()
    up = (lo + 1).clamp_(0, 2 * support_size)
    lo = lo.clamp_(0, 2 * support_size)
    sup.scatter_(1, lo.unsqueeze(1), lo_p.unsqueeze(1))
    sup.scatter_(1, up.unsqueeze(1), up_p.unsqueeze(1))
    return -(sup * F.log_softmax(logits, dim=1)).sum(1)
