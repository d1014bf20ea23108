"""Trainer step on the device that produced the trajectories.

`train_network_from_tensors` mirrors `v1/python/train_bridge.py:108-545` -- same keyword arguments, same metrics
dictionary -- with the loss assembly replaced by the fused HIP kernel (`train_loss.fused_policy_value_loss`) and the
samples taken where they are: a `TensorSelfPlayBatch` that is still resident in HBM is trained on without the
reference's CPU round trip.  The network forward / backward itself is PyTorch-ROCm autograd (MIOpen convolutions)
under AMP, Adam + warm-up + gradient clipping as in the reference.  `parallel_strategy="ddp"` uses
`torch.nn.parallel.DistributedDataParallel` over the initialised process group (RCCL), sharding rows `rank::world`
exactly like the reference.
"""
from __future__ import annotations

import os
import time
from contextlib import nullcontext
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn, optim

from .train_loss import fused_policy_value_loss
from .trajectory_buffer import TensorSelfPlayBatch


def _all_ranks_true(flag: bool, ddp: bool, device: torch.device) -> bool:
    if not ddp:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def train_network_from_tensors(model, samples: TensorSelfPlayBatch, *, batch_size: int = 512, epochs: int = 1,
                               lr: float = 1e-3, weight_decay: float = 1e-4, soft_label_alpha: float = 0.0,
                               anti_draw_penalty: float = 0.0, policy_draw_weight: float = 1.0, device: str = "cuda:0",
                               use_amp: bool = True, grad_clip_norm: float = 1.0, warmup_steps: int = 0,
                               parallel_devices: Optional[List[str]] = None, parallel_strategy: str = "none",
                               ddp_pre_sharded: bool = False, optimizer_state_path: Optional[str] = None
                               ) -> Tuple[Any, Dict[str, Any]]:
    if samples.num_samples <= 0:
        return model, {"epoch_stats": [], "num_samples": 0}
    strategy = {"dp": "none", "data_parallel": "none", "none": "none", "single": "none", "ddp": "ddp"}.get(
        str(parallel_strategy).strip().lower())
    if strategy is None:
        raise ValueError(f"Unsupported parallel_strategy={parallel_strategy!r}; expected one of: none, data_parallel, ddp.")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("liuzhou_amd trainer step needs a HIP device (no CPU path)")
    rank, world = 0, 1
    if strategy == "ddp":
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("parallel_strategy='ddp' requires torch.distributed to be initialized. Launch with torchrun.")
        dev = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
        torch.cuda.set_device(dev)
        rank, world = int(dist.get_rank()), int(dist.get_world_size())
    model.to(dev)
    model.train()
    train_model: nn.Module = model
    if strategy == "ddp":
        train_model = nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], output_device=dev.index,
                                                          broadcast_buffers=False, find_unused_parameters=False)
    global_n = int(samples.num_samples)
    t0 = time.perf_counter()
    sl = slice(rank, None, world) if (strategy == "ddp" and world > 1 and not ddp_pre_sharded) else slice(None)
    src = [getattr(samples, k)[sl] for k in ("state_tensors", "legal_masks", "policy_targets", "value_targets",
                                             "soft_value_targets")]
    shard_sec = time.perf_counter() - t0
    t0 = time.perf_counter()
    states = src[0].to(dev, non_blocking=True).to(torch.float32)
    masks = src[1].to(dev, non_blocking=True).to(torch.bool)
    policy = src[2].to(dev, non_blocking=True).to(torch.float32)
    values = src[3].to(dev, non_blocking=True).to(torch.float32).view(-1)
    soft = src[4].to(dev, non_blocking=True).to(torch.float32).view(-1)
    copy_sec = time.perf_counter() - t0
    # rows with a non-finite target are dropped (train_bridge.py:210-230)
    finite = torch.isfinite(values) & torch.isfinite(soft) & torch.isfinite(policy).all(dim=1) & \
        torch.isfinite(states.view(states.size(0), -1)).all(dim=1)
    filtered = int(finite.numel() - int(finite.sum().item()))
    if filtered:
        keep = torch.nonzero(finite).view(-1)
        states, masks, policy, values, soft = (t.index_select(0, keep) for t in (states, masks, policy, values, soft))
    n = int(states.shape[0])
    if n <= 0:
        if strategy == "ddp":
            raise RuntimeError("DDP received no local samples on one rank. Increase self-play samples or reduce world size.")
        return model, {"epoch_stats": [], "num_samples": 0}

    optimizer = optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    opt_loaded, opt_err = False, None
    if optimizer_state_path and os.path.exists(optimizer_state_path):
        try:
            optimizer.load_state_dict(torch.load(optimizer_state_path, map_location=dev))
            for pg in optimizer.param_groups:
                pg["lr"] = float(lr)
                pg["initial_lr"] = float(lr)
            opt_loaded = True
        except Exception as exc:   # fresh Adam, like the reference
            opt_err = repr(exc)
    amp = bool(use_amp)
    scaler = torch.amp.GradScaler("cuda", enabled=True) if amp else None
    alpha = float(max(0.0, min(1.0, soft_label_alpha)))
    draw_w = float(max(0.0, policy_draw_weight))
    bsz = max(1, int(batch_size))
    local_batches = (n + bsz - 1) // bsz
    synced = local_batches
    if strategy == "ddp" and world > 1:
        tok = torch.tensor([local_batches], dtype=torch.int64, device=dev)
        dist.all_reduce(tok, op=dist.ReduceOp.MIN)
        synced = max(0, int(tok.item()))
    dropped = max(0, n - synced * bsz)
    n_epochs = max(1, int(epochs))
    total_steps = synced * n_epochs
    warm = min(max(0, int(warmup_steps)), total_steps // 2)
    scheduler = torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda step: (step + 1) / max(1, warm) if (warm > 0 and step < warm) else 1.0)
    lr_start = float(optimizer.param_groups[0]["lr"])
    epoch_stats: List[Dict[str, Any]] = []
    first_batch_sec, first_done = 0.0, False
    for epoch in range(n_epochs):
        perm = torch.randperm(n, device=dev)
        acc = torch.zeros(9, dtype=torch.float64, device=dev)   # loss, policy*w, value, bucket, aux, seen, wsum, valid, |soft|
        mix_abs_sum, batches, skip_loss, skip_grad = 0.0, 0, 0, 0
        for step_idx in range(synced):
            start = step_idx * bsz
            if start >= n:
                break
            tb = time.perf_counter()
            idx = perm[start:min(start + bsz, n)]
            b_states, b_masks, b_policy = states.index_select(0, idx), masks.index_select(0, idx), policy.index_select(0, idx)
            b_values, b_soft = values.index_select(0, idx), soft.index_select(0, idx)
            optimizer.zero_grad(set_to_none=True)
            with (torch.amp.autocast("cuda", enabled=True) if amp else nullcontext()):
                lp1, lp2, lpm, vlogits = train_model(b_states)
            loss, parts = fused_policy_value_loss(lp1, lp2, lpm, vlogits, b_masks, b_policy, b_values, b_soft,
                                                  soft_label_alpha=alpha, anti_draw_penalty=float(anti_draw_penalty),
                                                  policy_draw_weight=draw_w)
            if not _all_ranks_true(bool(torch.isfinite(loss).item()), strategy == "ddp", dev):
                skip_loss += 1
                optimizer.zero_grad(set_to_none=True)
                continue
            if scaler is not None:
                scaler.scale(loss).backward()
                scaler.unscale_(optimizer)
            else:
                loss.backward()
            grads_ok = all(p.grad is None or bool(torch.isfinite(p.grad).all().item()) for p in model.parameters())
            if not _all_ranks_true(grads_ok, strategy == "ddp", dev):
                skip_grad += 1
                optimizer.zero_grad(set_to_none=True)
                if scaler is not None:
                    scaler.update()
                continue
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=float(grad_clip_norm))
            if scaler is not None:
                scaler.step(optimizer)
                scaler.update()
            else:
                optimizer.step()
            scheduler.step()
            cnt = float(idx.numel())
            draw = b_values.abs() < 1e-8
            wsum = torch.where(draw, draw_w, 1.0).sum()
            v_used = (torch.where(draw, torch.full_like(b_values, float(anti_draw_penalty)), b_values)
                      if abs(float(anti_draw_penalty)) > 1e-9 else b_values)
            mixed = ((1.0 - alpha) * v_used + alpha * b_soft).clamp(-1.0, 1.0)
            acc += torch.stack([loss.detach() * cnt, parts["policy_loss"] * wsum,
                                parts["bucket_value_loss"] * cnt, parts["bucket_value_loss"] * cnt,
                                parts["wdl_aux_loss"] * cnt, torch.tensor(cnt, device=dev), wsum,
                                (b_policy.sum(dim=1) > 1e-8).sum(), b_soft.abs().mean()]).to(torch.float64)
            mix_abs_sum += float(mixed.abs().mean().item())
            batches += 1
            if not first_done:
                first_batch_sec, first_done = time.perf_counter() - tb, True
        red = torch.cat([acc, torch.tensor([mix_abs_sum, float(batches), float(skip_loss), float(skip_grad)],
                                           dtype=torch.float64, device=dev)])
        if strategy == "ddp":
            dist.all_reduce(red, op=dist.ReduceOp.SUM)
        r = red.tolist()
        seen = int(round(r[5]))
        epoch_stats.append({
            "epoch": epoch + 1, "avg_loss": r[0] / max(1, seen), "avg_policy_loss": r[1] / max(1e-8, r[6]),
            "avg_value_loss": r[2] / max(1, seen), "avg_value_bucket_loss": r[3] / max(1, seen),
            "avg_wdl_aux_loss": r[4] / max(1, seen), "samples": seen, "valid_policy_samples": int(round(r[7])),
            "policy_weight_sum": r[6], "soft_alpha": alpha, "avg_soft_abs": r[8] / max(1, int(round(r[10]))),
            "avg_mix_abs": r[9] / max(1, int(round(r[10]))), "parallel_strategy": strategy, "ddp_world_size": world,
            "local_batch_count": int(local_batches), "synced_batch_count": int(synced),
            "dropped_samples_for_sync": int(dropped), "skipped_non_finite_loss_batches": int(round(r[11])),
            "skipped_non_finite_grad_batches": int(round(r[12])), "filtered_non_finite_samples": filtered})
    lr_final = float(optimizer.param_groups[0]["lr"])
    if optimizer_state_path and (strategy != "ddp" or rank == 0):
        try:
            torch.save(optimizer.state_dict(), optimizer_state_path)
        except Exception:
            pass
    return model, {
        "epoch_stats": epoch_stats, "num_samples": global_n, "num_samples_after_filter": n,
        "filtered_non_finite_samples": filtered, "parallel_strategy": strategy, "ddp_world_size": world,
        "local_batch_count": int(local_batches), "synced_batch_count": int(synced),
        "dropped_samples_for_sync": int(dropped), "optimizer_loaded": opt_loaded, "optimizer_load_error": opt_err,
        "optimizer_lr_start": lr_start, "optimizer_lr_final": lr_final, "device": str(dev),
        "device_fallback_count": 0, "device_fallback_reasons": [], "anti_draw_penalty": float(anti_draw_penalty),
        "wdl_aux_loss_weight": 0.0, "warmup_steps": int(warm), "total_train_steps": int(total_steps),
        "timing": {"cpu_shard_sec": float(shard_sec), "h2d_copy_sec": float(copy_sec),
                   "first_batch_sec": float(first_batch_sec), "ddp_pre_sharded": bool(ddp_pre_sharded)}}
