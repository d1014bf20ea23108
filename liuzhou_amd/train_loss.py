"""Fused policy + bucketed-value training loss on the HIP library.

`fused_policy_value_loss(...)` is the drop-in for the loss assembly of `v1/python/train_bridge.py:330-375`
(`build_combined_logits` -> `masked_log_softmax` -> `batched_policy_loss`, two-hot bucket cross entropy): one kernel
computes the per-sample terms and the gradients with respect to the four head outputs; autograd only multiplies
them by the incoming scalar gradient (which carries the AMP loss scale).  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib as L


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp1, lp2, lpm, vlogits, legal_mask, policy_target, value_target, soft_target, alpha, anti_draw,
                draw_weight):
        dev = lp1.device
        if dev.type != "cuda":
            L.require_hip(lp1, "fused_policy_value_loss")
        B = int(lp1.shape[0])
        f32 = lambda t: t.detach().reshape(B, -1).to(torch.float32).contiguous()
        a1, a2, a3, av = f32(lp1), f32(lp2), f32(lpm), f32(vlogits)
        mask = legal_mask.reshape(B, -1).contiguous()
        mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        tgt = policy_target.reshape(B, -1).to(torch.float32).contiguous()
        val = value_target.reshape(B).to(torch.float32).contiguous()
        soft = soft_target.reshape(B).to(torch.float32).contiguous()
        if a1.shape[1] != 36 or av.shape[1] != 101 or mask.shape[1] != 220 or tgt.shape[1] != 220:
            raise RuntimeError("fused loss expects 6x6 heads [B,36], 101 value bins and 220 actions")
        wsum = torch.where(val.abs() < 1e-8, float(draw_weight), 1.0).to(torch.float32).sum().reshape(1)
        terms = torch.empty((B, 4), dtype=torch.float32, device=dev)
        g1, g2, g3 = (torch.empty((B, 36), dtype=torch.float32, device=dev) for _ in range(3))
        gv = torch.empty((B, 101), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().lz_policy_value_loss_fwd_bwd(
                L.ptr(a1), L.ptr(a2), L.ptr(a3), L.ptr(av), L.ptr(mask), L.ptr(tgt), L.ptr(val), L.ptr(soft), L.i64(B),
                C.c_float(float(alpha)), C.c_float(float(anti_draw)), C.c_float(float(draw_weight)), L.ptr(wsum),
                C.c_float(1.0), L.ptr(terms), L.ptr(g1), L.ptr(g2), L.ptr(g3), L.ptr(gv), L.stream_ptr(dev)),
                "policy_value_loss_fwd_bwd")
        policy_loss = (terms[:, 0] * terms[:, 1]).sum() / (wsum[0] + 1e-8)
        bucket_loss = terms[:, 2].mean()
        wdl_aux = terms[:, 3].mean()
        ctx.save_for_backward(g1, g2, g3, gv)
        ctx.shapes = (lp1.shape, lp2.shape, lpm.shape, vlogits.shape)
        ctx.dtypes = (lp1.dtype, lp2.dtype, lpm.dtype, vlogits.dtype)
        ctx.mark_non_differentiable(policy_loss, bucket_loss, wdl_aux)
        return policy_loss + bucket_loss, policy_loss, bucket_loss, wdl_aux

    @staticmethod
    def backward(ctx, grad_loss, _gp, _gb, _ga):
        g1, g2, g3, gv = ctx.saved_tensors
        outs = []
        for g, shape, dt in zip((g1, g2, g3, gv), ctx.shapes, ctx.dtypes):
            outs.append((g * grad_loss).to(dt).reshape(shape))
        return (*outs, None, None, None, None, None, None, None)


def fused_policy_value_loss(log_p1: torch.Tensor, log_p2: torch.Tensor, log_pmc: torch.Tensor,
                            value_logits: torch.Tensor, legal_mask: torch.Tensor, policy_target: torch.Tensor,
                            value_target: torch.Tensor, soft_value_target: torch.Tensor, *,
                            soft_label_alpha: float = 0.0, anti_draw_penalty: float = 0.0,
                            policy_draw_weight: float = 1.0) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """loss = weighted policy KL + mean two-hot bucket cross entropy; parts = policy_loss, bucket_value_loss,
    wdl_aux_loss (reported only: its weight is 0 in the reference, train_bridge.py:25)."""
    L.require_hip(log_p1, "fused_policy_value_loss")
    alpha = float(max(0.0, min(1.0, soft_label_alpha)))
    loss, pol, bucket, aux = _FusedLoss.apply(log_p1, log_p2, log_pmc, value_logits, legal_mask, policy_target,
                                              value_target, soft_value_target, alpha, float(anti_draw_penalty),
                                              float(max(0.0, policy_draw_weight)))
    return loss, {"policy_loss": pol, "bucket_value_loss": bucket, "wdl_aux_loss": aux}
