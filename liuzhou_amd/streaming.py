"""Streaming batches straight from self-play shard files (v1/python/streaming_dataset.py:140-285): an
`IterableDataset` that deals the shards of a manifest to DataLoader workers, keeps each loaded shard in the worker's
memory for later epochs, sub-samples replay shards to their budget and emits shuffled tensor batches.
Shard discovery / budgets: `self_play_stage.resolve_shard_specs`."""
from __future__ import annotations

import random
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch
import torch.utils.data

from .self_play_stage import TENSOR_KEYS, ShardSpec, _batch_from_obj, _load

Batch5 = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]
_shard_cache: Dict[str, Batch5] = {}          # per worker process


def _load_shard(path: str) -> Optional[Batch5]:
    try:
        b = _batch_from_obj(_load(path), path)
    except Exception:
        return None
    return tuple(getattr(b, k) for k in TENSOR_KEYS)  # type: ignore[return-value]


class StreamingSelfPlayDataset(torch.utils.data.IterableDataset):
    def __init__(self, shard_specs: Sequence[ShardSpec], *, batch_size: int, epoch_seed: int = 0) -> None:
        super().__init__()
        self.specs = list(shard_specs)
        self.batch_size = max(1, int(batch_size))
        self.epoch_seed = int(epoch_seed)
        self.epoch = 0

    def __iter__(self) -> Iterator[Batch5]:
        info = torch.utils.data.get_worker_info()
        wid, nw = (info.id, info.num_workers) if info is not None else (0, 1)
        mine = [s for i, s in enumerate(self.specs) if i % nw == wid]
        if not mine:
            return
        epoch, self.epoch = self.epoch, self.epoch + 1
        random.Random(self.epoch_seed * 10007 + wid * 31 + epoch).shuffle(mine)
        for spec in mine:
            tensors = _shard_cache.get(spec.path)
            if tensors is None:
                tensors = _load_shard(spec.path)
                if tensors is None:
                    continue
                _shard_cache[spec.path] = tensors
            n = int(tensors[0].shape[0])
            if n <= 0:
                continue
            if 0 < spec.sample_budget < n:
                pick = torch.randperm(n)[: spec.sample_budget]
                tensors = tuple(t.index_select(0, pick) for t in tensors)
                n = spec.sample_budget
            perm = torch.randperm(n)
            for start in range(0, n, self.batch_size):
                idx = perm[start:start + self.batch_size]
                yield tuple(t.index_select(0, idx) for t in tensors)


def build_streaming_dataloader(shard_specs: Sequence[ShardSpec], *, batch_size: int, num_workers: int = 1,
                               epoch_seed: int = 0, pin_memory: bool = True, prefetch_factor: int = 2
                               ) -> torch.utils.data.DataLoader:
    workers = min(int(num_workers), len(shard_specs)) if shard_specs else 0
    ds = StreamingSelfPlayDataset(shard_specs, batch_size=batch_size, epoch_seed=epoch_seed)
    return torch.utils.data.DataLoader(ds, batch_size=None, num_workers=workers, pin_memory=pin_memory,
                                       prefetch_factor=prefetch_factor if workers > 0 else None,
                                       persistent_workers=workers > 0)
