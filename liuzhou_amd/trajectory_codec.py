"""Exact 360-byte records for trajectory rows (7.5x smaller than the 2 692-byte 5-tensor row): the wire format of
`distributed.gather_trajectories` between GPUs.  HIP only; the on-disk shards keep the
reference's tensor payload."""
from __future__ import annotations

import torch

from . import _lib as L
from .trajectory_buffer import TensorSelfPlayBatch

RECORD_BYTES = 360


def pack_batch(batch: TensorSelfPlayBatch, return_bad: bool = False):
    """TensorSelfPlayBatch on a HIP device -> uint8[n, 360].  `return_bad`: do not read the count of non-representable
    rows back (no host synchronisation, nothing raised here) but return it as a device int32[1] next to the records,
    for callers that must decide collectively (distributed._gather_compact)."""
    L.require_hip(batch.state_tensors, "pack_trajectory_rows")
    dev = batch.state_tensors.device
    n = int(batch.num_samples)
    out = torch.empty((n, RECORD_BYTES), dtype=torch.uint8, device=dev)
    bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    planes = batch.state_tensors.to(torch.float32).contiguous()
    legal = batch.legal_masks.contiguous()
    legal = legal.view(torch.uint8) if legal.dtype == torch.bool else legal.to(torch.uint8)
    pol = batch.policy_targets.to(torch.float32).contiguous()
    val = batch.value_targets.to(torch.float32).reshape(-1).contiguous()
    soft = batch.soft_value_targets.to(torch.float32).reshape(-1).contiguous()
    with torch.cuda.device(dev):
        L.check(L.lib().lz_pack_trajectory_rows(L.ptr(planes), L.ptr(legal), L.ptr(pol), L.ptr(val), L.ptr(soft), L.i64(n),
                                                L.ptr(out), L.ptr(bad), L.stream_ptr(dev)), "pack_trajectory_rows")
    if return_bad:
        return out, bad
    if n and int(bad.item()) != 0:
        raise RuntimeError(f"{int(bad.item())} trajectory rows are not representable as compact records "
                           "(planes not 0/1, policy mass off the legal set, or more than 72 legal actions)")
    return out


def unpack_records(records: torch.Tensor) -> TensorSelfPlayBatch:
    """uint8[n, 360] on a HIP device -> the five tensors of the trajectory contract."""
    L.require_hip(records, "unpack_trajectory_rows")
    if records.dtype != torch.uint8 or records.dim() != 2 or int(records.shape[1]) != RECORD_BYTES:
        raise ValueError(f"expected uint8[n, {RECORD_BYTES}] records, got {records.dtype} {tuple(records.shape)}")
    dev = records.device
    n = int(records.shape[0])
    rec = records.contiguous()
    planes = torch.empty((n, 11, 6, 6), dtype=torch.float32, device=dev)
    legal = torch.empty((n, 220), dtype=torch.uint8, device=dev)
    pol = torch.empty((n, 220), dtype=torch.float32, device=dev)
    val = torch.empty((n,), dtype=torch.float32, device=dev)
    soft = torch.empty((n,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().lz_unpack_trajectory_rows(L.ptr(rec), L.i64(n), L.ptr(planes), L.ptr(legal), L.ptr(pol),
                                                  L.ptr(val), L.ptr(soft), L.stream_ptr(dev)), "unpack_trajectory_rows")
    return TensorSelfPlayBatch(state_tensors=planes, legal_masks=legal.view(torch.bool), policy_targets=pol,
                               value_targets=val, soft_value_targets=soft)
