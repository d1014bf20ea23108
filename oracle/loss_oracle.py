"""CPU ORACLE of the training loss (test infrastructure only): plain fp32 PyTorch restatement of
  v1/python/train_bridge.py:330-375 (loss assembly), src/policy_batch.py:95-189 (combined logits, masked
  log-softmax, weighted policy KL), src/neural_network.py:163-198 (scalar -> WDL / two-hot buckets),
  v1/python/train_bridge.py:30-41 (bucket logits -> WDL probabilities).
Pinned by tests/golden/g11_loss.npz (values and autograd gradients of the reference's own functions)."""
from __future__ import annotations

from typing import Dict

import torch

DIRS = ((-1, 0), (1, 0), (0, -1), (0, 1))


def combined_logits(lp1: torch.Tensor, lp2: torch.Tensor, lpm: torch.Tensor) -> torch.Tensor:
    B = lp1.shape[0]
    out = torch.zeros((B, 220), dtype=lp1.dtype)
    out[:, :36] = lp1
    for cell in range(36):
        r, c = divmod(cell, 6)
        for d, (dr, dc) in enumerate(DIRS):
            nr, nc = r + dr, c + dc
            if 0 <= nr < 6 and 0 <= nc < 6:
                out[:, 36 + cell * 4 + d] = lp2[:, cell] + lp1[:, nr * 6 + nc]
            else:
                out[:, 36 + cell * 4 + d] = float("-inf")
    out[:, 180:216] = lpm
    return out


def policy_value_loss(lp1, lp2, lpm, value_logits, legal_mask, policy_target, value_target, soft_target, *,
                      soft_label_alpha: float = 0.0, anti_draw_penalty: float = 0.0,
                      policy_draw_weight: float = 1.0) -> Dict[str, torch.Tensor]:
    """All inputs fp32 CPU tensors (heads may require grad).  Returns loss and its parts."""
    lp1, lp2, lpm = lp1.float(), lp2.float(), lpm.float()
    comb = combined_logits(lp1, lp2, lpm)
    neg = torch.full_like(comb, float("-inf"))
    masked = torch.where(legal_mask, comb, neg)
    lse = torch.logsumexp(masked, dim=1, keepdim=True)
    ok = legal_mask.any(dim=1, keepdim=True) & torch.isfinite(lse)
    lse = torch.where(ok, lse, torch.zeros_like(lse))
    logp = torch.where(legal_mask, masked - lse, torch.zeros_like(comb))
    logp = torch.where(torch.isfinite(logp), logp, torch.full_like(logp, -50.0))
    ce = -(policy_target * logp.clamp(min=-50.0)).sum(dim=1)
    ent = -(policy_target * policy_target.clamp(min=1e-8).log()).sum(dim=1)
    kl = ce - ent
    raw = value_target.view(-1)
    draw = raw.abs() < 1e-8
    w = torch.where(draw, torch.full_like(raw, float(policy_draw_weight)), torch.ones_like(raw))
    policy_loss = (kl * w).sum() / (w.sum() + 1e-8)

    v = raw.clone()
    if abs(float(anti_draw_penalty)) > 1e-9:
        v[draw] = float(anti_draw_penalty)
    a = float(max(0.0, min(1.0, soft_label_alpha)))
    mixed = ((1.0 - a) * v + a * soft_target.view(-1)).clamp(-1.0, 1.0)
    bins = value_logits.shape[1]
    step = 2.0 / (bins - 1)
    u = (mixed + 1.0) / step
    lo = torch.floor(u).long().clamp(0, bins - 1)
    hi = (lo + 1).clamp(0, bins - 1)
    frac = (u - lo.float()).clamp(0.0, 1.0)
    frac = torch.where(hi == lo, torch.zeros_like(frac), frac)
    tgt = torch.zeros((raw.shape[0], bins), dtype=torch.float32)
    tgt.scatter_add_(1, lo.view(-1, 1), (1.0 - frac).view(-1, 1))
    tgt.scatter_add_(1, hi.view(-1, 1), frac.view(-1, 1))
    vlog = torch.log_softmax(value_logits.float(), dim=1)
    ce_v = -(tgt * vlog).sum(dim=1)
    bucket_loss = ce_v.mean()

    probs = torch.softmax(value_logits.float(), dim=1)
    centers = torch.linspace(-1.0, 1.0, steps=bins)
    pw, pl = probs[:, centers > 1e-8].sum(1), probs[:, centers < -1e-8].sum(1)
    pd = probs[:, (centers.abs() <= 1e-8)].sum(1)
    wdl = torch.stack([pw, pd, pl], dim=1)
    wdl = wdl / wdl.sum(dim=1, keepdim=True).clamp_min(1e-8)
    tw, tl = raw.clamp(min=0.0), (-raw).clamp(min=0.0)
    wdl_t = torch.stack([tw, (1.0 - tw - tl).clamp(min=0.0), tl], dim=1)
    aux = -(wdl_t * torch.log(wdl.clamp_min(1e-8))).sum(dim=1)
    return dict(loss=policy_loss + bucket_loss, policy_loss=policy_loss, bucket_loss=bucket_loss, kl=kl, weight=w,
                ce_value=ce_v, wdl_aux=aux, mixed=mixed)
