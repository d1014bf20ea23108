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


class _Stepper:
    """One optimisation step on a batch (forward under AMP, fused loss, backward, finite checks, clip, Adam) and the
    running sums of an epoch -- shared by the in-memory and the streaming trainer."""

    def __init__(self, model, train_model, optimizer, scheduler, scaler, *, amp: bool, alpha: float, anti_draw: float,
                 draw_w: float, grad_clip_norm: float, ddp: bool, dev: torch.device) -> None:
        self.model, self.train_model, self.optimizer, self.scheduler, self.scaler = model, train_model, optimizer, scheduler, scaler
        self.amp, self.alpha, self.anti_draw, self.draw_w = amp, alpha, anti_draw, draw_w
        self.clip, self.ddp, self.dev = float(grad_clip_norm), ddp, dev
        self._nominal_rows: Optional[int] = None
        self._row_hist: Dict[int, int] = {}               # batch sizes seen so far
        self._conv_counts = [0, 0]                        # steps in [search-once, immediate] mode (reported per epoch)
        self._conv_logged = False
        self.reset()

    def reset(self) -> None:
        # loss, policy*w, value, bucket, aux, seen, wsum, valid, |soft|
        self.acc = torch.zeros(9, dtype=torch.float64, device=self.dev)
        self.mix_abs_dev = torch.zeros((), dtype=torch.float64, device=self.dev)
        self.batches, self.skip_loss, self.skip_grad = 0, 0, 0

    def _conv_mode(self, rows: int):
        """MIOpen searches for convolution kernels the first time it sees a shape: 12 - 20 s for the nominal batch of this
        network (once per process) and 2 - 11 s for EVERY new last-batch size -- and an iteration of the staged loop has a
        different sample count, hence a different remainder, every time (measured: `profiles/r03_train.md`).  Immediate
        mode (`torch.backends.miopen.immediate`) picks a kernel without searching: same samples/s for 10x128, 11 % lower
        for 6x64.  LZ_TRAIN_MIOPEN = auto (default: the nominal batch size searches once, any other size runs immediate),
        immediate (never search) or find (PyTorch's default behaviour)."""
        mode = os.environ.get("LZ_TRAIN_MIOPEN", "auto").strip().lower()
        # nominal = the row count that is worth one kernel search.  The callers set it to their batch size; when the
        # loader's batches regularly have another size (its own batch size, rows dropped by the non-finite filter), the
        # size seen most often takes over after its third occurrence instead of every step running immediate mode
        rows = int(rows)
        self._row_hist[rows] = self._row_hist.get(rows, 0) + 1
        if self._nominal_rows is None:
            self._nominal_rows = rows
        elif rows != self._nominal_rows and self._row_hist[rows] >= 3 and \
                self._row_hist[rows] > self._row_hist.get(self._nominal_rows, 0):
            self._nominal_rows = rows
            self._conv_logged = False
        imm = mode == "immediate" or (mode == "auto" and int(rows) != self._nominal_rows)
        self._conv_counts[1 if imm else 0] += 1
        if not self._conv_logged:
            self._conv_logged = True
            print(f"[liuzhou_amd.train] MIOpen mode {mode}: batches of {self._nominal_rows} rows search once, every other "
                  f"size runs immediate mode", flush=True)
        return torch.backends.miopen.flags(immediate=True) if (imm and hasattr(torch.backends, "miopen")) else nullcontext()

    def step(self, b_states, b_masks, b_policy, b_values, b_soft) -> bool:
        with self._conv_mode(int(b_states.shape[0])):
            return self._step(b_states, b_masks, b_policy, b_values, b_soft)

    def _step(self, b_states, b_masks, b_policy, b_values, b_soft) -> bool:
        opt, scaler, dev = self.optimizer, self.scaler, self.dev
        opt.zero_grad(set_to_none=True)
        with (torch.amp.autocast("cuda", enabled=True) if self.amp else nullcontext()):
            lp1, lp2, lpm, vlogits = self.train_model(b_states)
        loss, parts = fused_policy_value_loss(lp1, lp2, lpm, vlogits, b_masks, b_policy, b_values, b_soft,
                                              soft_label_alpha=self.alpha, anti_draw_penalty=self.anti_draw,
                                              policy_draw_weight=self.draw_w)
        # ONE host read per step for both finite checks of the reference (train_bridge.py:388-420 reads the loss flag and
        # then one flag per parameter tensor: ~130 device round trips per step).  The backward runs before the read; a
        # non-finite loss makes non-finite gradients, which are thrown away below exactly as if backward had not run
        # (no unscale_, no scaler.update(): the scaler never saw the step).  Finiteness is the same before and after
        # unscale_ (a division by the finite scale), so the gradient flag is taken on the scaled gradients.
        if scaler is not None:
            scaler.scale(loss).backward()
        else:
            loss.backward()
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        g_ok = torch.stack([torch.isfinite(g).all() for g in grads]).all() if grads else torch.ones((), dtype=torch.bool, device=dev)
        loss_ok, grads_ok = (bool(v) for v in torch.stack([torch.isfinite(loss.detach()).all(), g_ok]).tolist())
        if not _all_ranks_true(loss_ok, self.ddp, dev):
            self.skip_loss += 1
            opt.zero_grad(set_to_none=True)
            return False
        if scaler is not None:
            scaler.unscale_(opt)
        if not _all_ranks_true(grads_ok, self.ddp, dev):
            self.skip_grad += 1
            opt.zero_grad(set_to_none=True)
            if scaler is not None:
                scaler.update()
            return False
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.clip)
        if scaler is not None:
            scaler.step(opt)
            scaler.update()
        else:
            opt.step()
        self.scheduler.step()
        cnt = float(b_values.numel())
        draw = b_values.abs() < 1e-8
        wsum = torch.where(draw, self.draw_w, 1.0).sum()
        v_used = (torch.where(draw, torch.full_like(b_values, self.anti_draw), b_values)
                  if abs(self.anti_draw) > 1e-9 else b_values)
        mixed = ((1.0 - self.alpha) * v_used + self.alpha * b_soft).clamp(-1.0, 1.0)
        self.acc += torch.stack([loss.detach() * cnt, parts["policy_loss"] * wsum, parts["bucket_value_loss"] * cnt,
                                 parts["bucket_value_loss"] * cnt, parts["wdl_aux_loss"] * cnt,
                                 torch.tensor(cnt, device=dev), wsum, (b_policy.sum(dim=1) > 1e-8).sum(),
                                 b_soft.abs().mean()]).to(torch.float64)
        self.mix_abs_dev += mixed.abs().mean().to(torch.float64)        # summed on the device, read once per epoch
        self.batches += 1
        return True

    def epoch_stats(self, epoch: int, extra: Dict[str, Any], more_sums: Optional[List[float]] = None):
        red = torch.cat([self.acc, self.mix_abs_dev.view(1),
                         torch.tensor([float(self.batches), float(self.skip_loss), float(self.skip_grad)] + list(more_sums or []),
                                      dtype=torch.float64, device=self.dev)])
        if self.ddp:
            dist.all_reduce(red, op=dist.ReduceOp.SUM)
        r = red.tolist()
        seen = int(round(r[5]))
        stats = {"epoch": epoch, "avg_loss": r[0] / max(1, seen), "avg_policy_loss": r[1] / max(1e-8, r[6]),
                 "avg_value_loss": r[2] / max(1, seen), "avg_value_bucket_loss": r[3] / max(1, seen),
                 "avg_wdl_aux_loss": r[4] / max(1, seen), "samples": seen, "valid_policy_samples": int(round(r[7])),
                 "policy_weight_sum": r[6], "soft_alpha": self.alpha, "avg_soft_abs": r[8] / max(1, int(round(r[10]))),
                 "avg_mix_abs": r[9] / max(1, int(round(r[10]))),
                 "skipped_non_finite_loss_batches": int(round(r[11])), "skipped_non_finite_grad_batches": int(round(r[12])),
                 "miopen_nominal_rows": self._nominal_rows, "miopen_immediate_steps": int(self._conv_counts[1]),
                 "miopen_search_mode_steps": int(self._conv_counts[0])}
        stats.update(extra)
        return stats, r[13:]


def _make_optimizer(model, lr, weight_decay, optimizer_state_path, dev):
    optimizer = optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    loaded, err = False, None
    if optimizer_state_path and os.path.exists(optimizer_state_path):
        try:
            optimizer.load_state_dict(torch.load(optimizer_state_path, map_location=dev))
            for pg in optimizer.param_groups:
                pg["lr"] = float(lr)
                pg["initial_lr"] = float(lr)
            loaded = True
        except Exception as exc:   # fresh Adam, like the reference
            err = repr(exc)
    return optimizer, loaded, err


def _resolve_strategy(parallel_strategy: str, device: str):
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
    return strategy, dev, rank, world


def _wrap(model, strategy, dev):
    model.to(dev)
    model.train()
    if strategy == "ddp":
        return nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], output_device=dev.index,
                                                   broadcast_buffers=False, find_unused_parameters=False)
    return model


def train_network_from_tensors(model, samples: TensorSelfPlayBatch, *, batch_size: int = 512, epochs: int = 1,
                               lr: float = 1e-3, weight_decay: float = 1e-4, soft_label_alpha: float = 0.0,
                               anti_draw_penalty: float = 0.0, policy_draw_weight: float = 1.0, device: str = "cuda:0",
                               use_amp: bool = True, grad_clip_norm: float = 1.0, warmup_steps: int = 0,
                               parallel_devices: Optional[List[str]] = None, parallel_strategy: str = "none",
                               ddp_pre_sharded: bool = False, optimizer_state_path: Optional[str] = None
                               ) -> Tuple[Any, Dict[str, Any]]:
    if samples.num_samples <= 0:
        return model, {"epoch_stats": [], "num_samples": 0}
    strategy, dev, rank, world = _resolve_strategy(parallel_strategy, device)
    train_model = _wrap(model, strategy, dev)
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

    optimizer, opt_loaded, opt_err = _make_optimizer(model, lr, weight_decay, optimizer_state_path, dev)
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
    stepper = _Stepper(model, train_model, optimizer, scheduler, scaler, amp=amp, alpha=alpha,
                       anti_draw=float(anti_draw_penalty), draw_w=draw_w, grad_clip_norm=grad_clip_norm,
                       ddp=strategy == "ddp", dev=dev)
    stepper._nominal_rows = bsz                       # the one batch size worth a MIOpen kernel search (see _conv_mode)
    for epoch in range(n_epochs):
        perm = torch.randperm(n, device=dev)
        stepper.reset()
        for step_idx in range(synced):
            start = step_idx * bsz
            if start >= n:
                break
            tb = time.perf_counter()
            idx = perm[start:min(start + bsz, n)]
            stepper.step(states.index_select(0, idx), masks.index_select(0, idx), policy.index_select(0, idx),
                         values.index_select(0, idx), soft.index_select(0, idx))
            if not first_done:
                first_batch_sec, first_done = time.perf_counter() - tb, True
        st, _ = stepper.epoch_stats(epoch + 1, {"parallel_strategy": strategy, "ddp_world_size": world,
                                                "local_batch_count": int(local_batches), "synced_batch_count": int(synced),
                                                "dropped_samples_for_sync": int(dropped),
                                                "filtered_non_finite_samples": filtered})
        epoch_stats.append(st)
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


def train_network_streaming(model, dataloader, *, total_samples: int, batch_size: int = 512, epochs: int = 1,
                            lr: float = 1e-3, weight_decay: float = 1e-4, soft_label_alpha: float = 0.0,
                            anti_draw_penalty: float = 0.0, policy_draw_weight: float = 1.0, device: str = "cuda:0",
                            use_amp: bool = True, grad_clip_norm: float = 1.0, warmup_steps: int = 0,
                            parallel_devices: Optional[List[str]] = None, parallel_strategy: str = "none",
                            optimizer_state_path: Optional[str] = None, streaming_workers: int = 8
                            ) -> Tuple[Any, Dict[str, Any]]:
    """Train from an iterable of (states, masks, policy, values, soft) batches -- `streaming.build_streaming_dataloader`
    over the shards of a self-play manifest -- mirroring `v1/python/train_bridge.py:547-900`: the number of steps per
    epoch is fixed up front from `total_samples` (synchronised with MIN over DDP ranks), an exhausted loader turns the
    remaining steps into no-ops that still take part in the rank votes, non-finite rows are dropped per batch."""
    strategy, dev, rank, world = _resolve_strategy(parallel_strategy, device)
    train_model = _wrap(model, strategy, dev)
    bsz = max(1, int(batch_size))
    est = max(1, (int(total_samples) + bsz - 1) // bsz)
    if strategy == "ddp" and world > 1:
        tok = torch.tensor([est], dtype=torch.int64, device=dev)
        dist.all_reduce(tok, op=dist.ReduceOp.MIN)
        est = max(1, int(tok.item()))
    n_epochs = max(1, int(epochs))
    total_steps = est * n_epochs
    warm = min(max(0, int(warmup_steps)), total_steps // 2)
    optimizer, opt_loaded, opt_err = _make_optimizer(model, lr, weight_decay, optimizer_state_path, dev)
    amp = bool(use_amp)
    scaler = torch.amp.GradScaler("cuda", enabled=True) if amp else None
    scheduler = torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda step: (step + 1) / max(1, warm) if (warm > 0 and step < warm) else 1.0)
    lr_start = float(optimizer.param_groups[0]["lr"])
    alpha = float(max(0.0, min(1.0, soft_label_alpha)))
    stepper = _Stepper(model, train_model, optimizer, scheduler, scaler, amp=amp, alpha=alpha,
                       anti_draw=float(anti_draw_penalty), draw_w=float(max(0.0, policy_draw_weight)),
                       grad_clip_norm=grad_clip_norm, ddp=strategy == "ddp", dev=dev)
    stepper._nominal_rows = bsz                       # the one batch size worth a MIOpen kernel search (see _conv_mode)
    epoch_stats: List[Dict[str, Any]] = []
    first_batch_sec, first_done = 0.0, False
    total_filtered, seen_all = 0, 0
    for epoch in range(n_epochs):
        stepper.reset()
        it = iter(dataloader)
        exhausted, exhausted_steps, batches = False, 0, 0
        for _ in range(est):
            tb = time.perf_counter()
            batch = None
            if not exhausted:
                try:
                    batch = next(it)
                except StopIteration:
                    exhausted = True
            if batch is None:
                exhausted_steps += 1
                _all_ranks_true(False, strategy == "ddp", dev)          # keep the rank votes aligned
                batches += 1
                continue
            b_states = batch[0].to(dev, non_blocking=True).float()
            b_masks = batch[1].to(dev, non_blocking=True).bool()
            b_policy = batch[2].to(dev, non_blocking=True).float()
            b_values = batch[3].to(dev, non_blocking=True).float().view(-1)
            b_soft = batch[4].to(dev, non_blocking=True).float().view(-1)
            finite = torch.isfinite(b_values) & torch.isfinite(b_soft) & torch.isfinite(b_policy).all(dim=1) & \
                torch.isfinite(b_states.view(b_states.size(0), -1)).all(dim=1)
            n_bad = int((~finite).sum().item())
            if n_bad:
                total_filtered += n_bad
                keep = torch.nonzero(finite).view(-1)
                if int(keep.numel()) == 0:
                    _all_ranks_true(False, strategy == "ddp", dev)
                    batches += 1
                    continue
                b_states, b_masks, b_policy, b_values, b_soft = (t.index_select(0, keep) for t in
                                                                 (b_states, b_masks, b_policy, b_values, b_soft))
            stepper.step(b_states, b_masks, b_policy, b_values, b_soft)
            batches += 1
            if not first_done:
                first_batch_sec, first_done = time.perf_counter() - tb, True
        st, more = stepper.epoch_stats(epoch + 1, {"parallel_strategy": strategy, "ddp_world_size": world,
                                                   "batches_this_epoch": int(batches)}, [float(exhausted_steps)])
        st["dataloader_exhausted_steps"] = int(round(more[0])) if more else int(exhausted_steps)
        seen_all += st["samples"]
        epoch_stats.append(st)
    lr_final = float(optimizer.param_groups[0]["lr"])
    if optimizer_state_path and (strategy != "ddp" or rank == 0):
        try:
            torch.save(optimizer.state_dict(), optimizer_state_path)
        except Exception:
            pass
    return model, {
        "epoch_stats": epoch_stats, "num_samples": int(total_samples), "num_samples_seen": int(seen_all),
        "filtered_non_finite_samples": int(total_filtered), "parallel_strategy": strategy, "ddp_world_size": world,
        "est_batches_per_epoch": int(est), "optimizer_loaded": opt_loaded, "optimizer_load_error": opt_err,
        "optimizer_lr_start": lr_start, "optimizer_lr_final": lr_final, "device": str(dev), "device_fallback_count": 0,
        "device_fallback_reasons": [], "anti_draw_penalty": float(anti_draw_penalty), "wdl_aux_loss_weight": 0.0,
        "warmup_steps": int(warm), "total_train_steps": int(total_steps), "streaming": True,
        "streaming_workers": int(streaming_workers), "timing": {"first_batch_sec": float(first_batch_sec)}}
