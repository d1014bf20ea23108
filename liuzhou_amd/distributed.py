"""Sharding self-play over the GPUs of one node + the once-per-iteration exchanges.

Games are independent, so ranks never talk while playing (v1/train.py:932-1020 shards by files; here one
process per GPU under torch.distributed).  Two collectives exist, both outside the hot loop:
  * gather_trajectories(): variable-length trajectory batches -> rank 0 (the trainer).  Counts are
    exchanged first, then every rank sends its rows to rank 0.  On RCCL over xGMI a gather-to-one runs
    as direct peer writes (each sender has its own link into the destination), so it is implemented with
    batched isend/irecv (ncclSend/ncclRecv groups) rather than a ring all-gather that would be bound by a
    single 153 GB/s link; an `all_gather` fallback exists for backends without P2P ops.
  * broadcast_checkpoint(): the state_dict (12 MB for 10x128) rank 0 -> all.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .trajectory_buffer import TensorSelfPlayBatch

_FIELDS = ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")
EXPANDED_ROW_BYTES = 11 * 36 * 4 + 220 + 220 * 4 + 4 + 4            # one row of the five-tensor contract (2 692 B)


def split_games(total_games: int, parts: int) -> List[int]:
    """v1/train.py:129-135: base + 1 for the first `total % parts` ranks."""
    parts, total = max(1, int(parts)), max(0, int(total_games))
    base, rem = divmod(total, parts)
    return [base + (1 if i < rem else 0) for i in range(parts)]


def worker_seed(iteration_seed: int, worker_idx: int) -> int:
    """v1/train.py:998"""
    return int(iteration_seed) * 10007 + (int(worker_idx) + 1) * 9973


def _free_device_bytes(dev: torch.device) -> int:
    """Device memory an allocation could still get: free on the device + cached blocks torch can reuse; -1 = not a HIP
    device (the gloo rehearsal with host tensors: no check)."""
    if dev.type != "cuda":
        return -1
    free, _total = torch.cuda.mem_get_info(dev)
    return int(free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))


def _gather_compact(batch: TensorSelfPlayBatch, dst: int, group) -> Optional[TensorSelfPlayBatch]:
    """HIP path: rows travel as exact 360-byte records (trajectory_codec.py), 7.5x less xGMI traffic than the five
    tensors, packed / unpacked by one kernel each side.  The row count and the number of rows the record format cannot
    hold travel together in the one all-gather that precedes the transfers, so a bad row on any rank makes EVERY rank
    raise before a single send / receive is posted (no rank is left waiting in a collective)."""
    from .trajectory_codec import RECORD_BYTES, pack_batch, unpack_records
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = batch.state_tensors.device
    rec, bad = pack_batch(batch, return_bad=True)
    # the destination's free memory travels with the counts, so that EVERY rank can see that the expanded rows would
    # not fit and raise before anything is sent (a check on `dst` alone would leave the senders waiting in the next
    # collective until it times out -- ADVICE r04)
    mine = torch.stack([torch.tensor(batch.num_samples, dtype=torch.int64, device=dev), bad.to(torch.int64).view(()),
                        torch.tensor(_free_device_bytes(dev) if rank == dst else -1, dtype=torch.int64, device=dev)])
    if dist.get_backend(group) != "nccl":
        mine = mine.cpu()
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    every = torch.stack(every).cpu()
    counts = [int(c) for c in every[:, 0]]
    n_bad = [int(b) for b in every[:, 1]]
    if any(n_bad):
        raise RuntimeError(f"gather_trajectories: rows not representable as compact records on ranks "
                           f"{[r for r, b in enumerate(n_bad) if b]} (counts {n_bad}): planes not 0/1, policy mass off "
                           "the legal set, or more than 72 legal actions; every rank raises, nothing was sent")
    # the destination expands every rank's records to the 2 692-byte five-tensor rows next to whatever else lives on
    # its device (at C4 rank 0 still holds its own tree arenas): refuse -- on every rank -- before anything moves
    need, free_dst = sum(counts) * (EXPANDED_ROW_BYTES + RECORD_BYTES), int(every[dst, 2])
    if free_dst >= 0 and need > free_dst:
        raise RuntimeError(f"gather_trajectories: rank {dst} needs {need / 2**30:.1f} GiB to receive and expand "
                           f"{sum(counts)} rows but only {free_dst / 2**30:.1f} GiB of its device memory are free; release "
                           "the search engines (arenas) first or gather in several pieces; every rank raises, nothing was "
                           "sent")
    # RCCL moves device memory directly (peer writes over xGMI); any other backend (gloo in the tests) stages the
    # records through host memory -- same protocol, same kernels on both sides
    direct = dist.get_backend(group) == "nccl"
    wire_dev = dev if direct else torch.device("cpu")
    if rank != dst:
        if counts[rank] > 0:
            out = rec if direct else rec.cpu()
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, out, dst, group)]):
                req.wait()
        return None
    buf = torch.empty((sum(counts), RECORD_BYTES), dtype=torch.uint8, device=wire_dev)
    ops, start = [], 0
    for r in range(world):
        if r == dst:
            buf[start:start + counts[r]].copy_(rec)
        elif counts[r] > 0:
            ops.append(dist.P2POp(dist.irecv, buf[start:start + counts[r]], r, group))
        start += counts[r]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    buf = buf.to(dev)
    return unpack_records(buf)


def gather_trajectories(batch: TensorSelfPlayBatch, dst: int = 0, group=None,
                        compact: Optional[bool] = None, force: bool = False) -> Optional[TensorSelfPlayBatch]:
    """Concatenate every rank's samples on `dst` (rank order).  Returns None on the other ranks.
    `compact` (default: on for HIP tensors) sends 360-byte records instead of the five tensors.
    `force`: run the protocol (counts all-gather, pack / unpack) even in a group of one -- what a 1-GPU box can execute
    of the RCCL path (tests/test_gpu_distributed.py)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return batch
    if compact is None:
        compact = batch.state_tensors.is_cuda
    if compact:
        return _gather_compact(batch, dst, group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = batch.state_tensors.device
    n_local = torch.tensor([batch.num_samples], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    fields = {f: getattr(batch, f).contiguous() for f in _FIELDS}
    if rank == dst:
        out: Dict[str, torch.Tensor] = {}
        total = sum(counts)
        ops, views = [], {}
        for f, t in fields.items():
            buf = torch.empty((total, *t.shape[1:]), dtype=t.dtype, device=dev)
            start = 0
            for r in range(world):
                if r == dst:
                    buf[start:start + counts[r]].copy_(t)
                elif counts[r] > 0:
                    tgt = buf[start:start + counts[r]]
                    if tgt.dtype == torch.bool:                      # bool is sent as uint8
                        tmp = torch.empty(tgt.shape, dtype=torch.uint8, device=dev)
                        views[(f, r)] = (tmp, tgt)
                        ops.append(dist.P2POp(dist.irecv, tmp, r, group))
                    else:
                        ops.append(dist.P2POp(dist.irecv, tgt, r, group))
                start += counts[r]
            out[f] = buf
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for (f, r), (tmp, tgt) in views.items():
            tgt.copy_(tmp.to(torch.bool))
        return TensorSelfPlayBatch(*(out[f] for f in _FIELDS))
    ops = []
    if counts[rank] > 0:
        for f, t in fields.items():
            ops.append(dist.P2POp(dist.isend, t.to(torch.uint8) if t.dtype == torch.bool else t, dst, group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return None


def broadcast_checkpoint(model: torch.nn.Module, src: int = 0, group=None, force: bool = False) -> None:
    """Rank `src`'s parameters and buffers -> every rank (checkpoint hand-off each iteration, v1/train.py:978 hands a
    `model_state_cpu.pt` file to the workers).  One flat buffer per dtype (fp32 weights + the int64 BatchNorm counters:
    12 MB for 10x128 in two collectives) instead of one tiny broadcast per tensor."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
    for t in list(model.parameters()) + list(model.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    direct = dist.get_backend(group) == "nccl"
    for _dt, ts in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.reshape(-1) for t in ts])
        if flat.is_cuda and not direct:                               # gloo rehearsal of the RCCL path: through host memory
            host = flat.cpu()
            dist.broadcast(host, src=src, group=group)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
