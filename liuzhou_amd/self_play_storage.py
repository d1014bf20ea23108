"""Shard payload writer for self-play output (same on-disk contract as v1/python/self_play_storage.py:14-106:
`torch.save` dict with keys state_tensors, legal_masks, policy_targets, value_targets, soft_value_targets,
stats, metadata), so the reference trainer / streaming dataset consume our shards unchanged."""
from __future__ import annotations

import math
import os
from typing import Any, Dict, List, Tuple

import torch

from .trajectory_buffer import TensorSelfPlayBatch

_FIELDS = ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")


def split_counts(total: int, parts: int) -> List[int]:
    total, parts = max(0, int(total)), max(1, int(parts))
    if total == 0:
        return []
    parts = min(parts, total)
    base, extra = divmod(total, parts)
    return [base + (1 if i < extra else 0) for i in range(parts)]


def estimate_bytes_per_sample(samples: TensorSelfPlayBatch) -> int:
    n = max(1, samples.num_samples)
    total = 0
    for name in _FIELDS:
        t = getattr(samples, name)
        if t.numel() > 0:
            total += t.element_size() * (t.numel() // n)
    return max(1, int(total))


def plan_sample_ranges(*, total_samples: int, num_shards: int, target_samples_per_shard: int = 0,
                       chunk_target_bytes: int = 0, bytes_per_sample: int = 0) -> List[Tuple[int, int]]:
    total = int(total_samples)
    if total <= 0:
        return []
    shards = max(1, min(int(num_shards), total))
    target = max(0, int(target_samples_per_shard))
    if int(chunk_target_bytes) > 0:
        target = max(1, int(chunk_target_bytes) // max(1, int(bytes_per_sample)))
    if target > 0:
        shards = min(total, max(shards, int(math.ceil(total / float(target)))))
    out, start = [], 0
    for size in split_counts(total, shards):
        out.append((start, start + size))
        start += size
    return out


def _own_storage(t: torch.Tensor) -> torch.Tensor:
    """Host tensor that owns exactly its bytes.  `torch.save` serialises a tensor's WHOLE underlying storage, so a
    slice of a host batch would write (and every loader of the shard would pin) the full batch per chunk."""
    t = t.detach().to("cpu")
    if not t.is_contiguous() or t.untyped_storage().nbytes() > t.numel() * t.element_size():
        t = t.contiguous().clone()
    return t


def slice_batch_cpu(samples: TensorSelfPlayBatch, *, start: int, end: int) -> TensorSelfPlayBatch:
    return TensorSelfPlayBatch(*(_own_storage(getattr(samples, f)[int(start):int(end)]) for f in _FIELDS))


def save_self_play_payload(*, path: str, samples: TensorSelfPlayBatch, stats_payload: Dict[str, Any],
                           metadata: Dict[str, Any]) -> None:
    payload = {f: _own_storage(getattr(samples, f)) for f in _FIELDS}
    payload["stats"] = dict(stats_payload)
    payload["metadata"] = dict(metadata)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(payload, path)
