"""Self-play stage: one worker process per device, chunk files + one sharded manifest, and the loader side.

This is the caller of `run_self_play_worker` and the reader of what it wrote, so that the reference's training
stage (`v1/python/streaming_dataset.py:69-140`, `v1/train.py:1588-1735`) consumes our output unchanged and we
consume its output:

  run_self_play_stage      ~ v1/train.py:932-1172  (_run_self_play_multi_device_process_saved)
  merge_worker_manifests   ~ v1/train.py:1056-1153 (worker manifests -> `v1_sharded_manifest`)
  save_sharded / load_*    ~ v1/train.py:1486-1559, :1588-1735
  resolve_shard_specs      ~ v1/python/streaming_dataset.py:69-140

On-disk contract (torch.save dicts):
  chunk     {state_tensors f32[n,11,6,6], legal_masks bool[n,220], policy_targets f32[n,220], value_targets f32[n],
             soft_value_targets f32[n], stats {}, metadata {payload_format: "v1_sharded_shard", ...}}
  manifest  {payload_format: "v1_sharded_manifest", version 1, num_samples, num_shards, shard_files (relative to the
             manifest), shard_sizes, chunk_target_bytes, avg_bytes_per_sample, stats, metadata}
Games shard over devices with no communication while they play; the only exchange is this
file hand-off (or `distributed.gather_trajectories` when the trainer lives in the same job).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import shutil
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor, as_completed
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import torch

from .distributed import split_games, worker_seed
from .self_play_storage import estimate_bytes_per_sample, plan_sample_ranges, save_self_play_payload, slice_batch_cpu
from .self_play_types import SelfPlayV1Stats
from .self_play_worker import merge_self_play_stats, merge_target_summaries, summarize_scalar_targets
from .trajectory_buffer import TensorSelfPlayBatch

TENSOR_KEYS = ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")


def stats_from_payload(payload: Any) -> SelfPlayV1Stats:
    """`SelfPlayV1Stats.to_dict()` payload -> record (missing fields default to zero / empty)."""
    p = payload if isinstance(payload, dict) else {}
    num = lambda k, cast: cast(p.get(k, 0) or 0)
    dct = lambda k, cast: {str(a): cast(b) for a, b in p[k].items()} if isinstance(p.get(k), dict) else {}
    elapsed = max(1e-9, num("elapsed_sec", float))
    games, positions = num("num_games", int), num("num_positions", int)
    buckets = {str(d): 0 for d in range(-18, 19)}
    for k, v in dct("piece_delta_buckets", int).items():
        if k in buckets:
            buckets[k] = v
    return SelfPlayV1Stats(
        num_games=games, num_positions=positions, black_wins=num("black_wins", int), white_wins=num("white_wins", int),
        draws=num("draws", int), avg_game_length=num("avg_game_length", float), elapsed_sec=elapsed,
        positions_per_sec=float(p.get("positions_per_sec", positions / elapsed)),
        games_per_sec=float(p.get("games_per_sec", games / elapsed)),
        step_timing_ms=dct("step_timing_ms", float), step_timing_ratio=dct("step_timing_ratio", float),
        step_timing_calls=dct("step_timing_calls", int), mcts_counters=dct("mcts_counters", int),
        piece_delta_buckets=buckets,
        policy_target_audit=dict(p["policy_target_audit"]) if isinstance(p.get("policy_target_audit"), dict) else {},
        device=str(p.get("device", "")), fallback_count=num("fallback_count", int),
        fallback_reasons=tuple(str(x) for x in (p.get("fallback_reasons") or [])))


def _load(path: str) -> Any:
    return torch.load(path, map_location="cpu", weights_only=False)


def merge_worker_manifests(worker_manifest_paths: Sequence[str], *, output_path: str, chunk_target_bytes: int = 0,
                           target_samples_per_shard: int = 0, metadata_base: Optional[Dict[str, Any]] = None,
                           elapsed_sec: float = 0.0) -> Dict[str, Any]:
    """Fold the per-worker `v1_worker_chunk_manifest` files (in worker order) into one `v1_sharded_manifest`
    next to the chunk files and return it."""
    files: List[str] = []
    sizes: List[int] = []
    stats: List[SelfPlayV1Stats] = []
    summaries: Dict[str, List[Dict[str, Any]]] = {"value_target_summary": [], "soft_value_target_summary": [],
                                                  "mixed_value_target_summary": []}
    bps_num = bps_den = 0
    for path in worker_manifest_paths:
        wm = _load(path)
        if not isinstance(wm, dict) or str(wm.get("payload_format", "")).strip().lower() != "v1_worker_chunk_manifest":
            raise RuntimeError(f"Invalid worker manifest payload_format in {path}: "
                               f"{wm.get('payload_format') if isinstance(wm, dict) else type(wm)!r}")
        wf, ws = wm.get("shard_files"), wm.get("shard_sizes")
        if not isinstance(wf, list) or not isinstance(ws, list):
            raise RuntimeError(f"Worker manifest missing shard file lists: {path}")
        for i, entry in enumerate(wf):
            name = str(entry).strip()
            if name:
                files.append(name)
                sizes.append(int(ws[i]) if i < len(ws) else 0)
        stats.append(stats_from_payload(wm.get("stats", {})))
        for key, bucket in summaries.items():
            if isinstance(wm.get(key), dict):
                bucket.append(wm[key])
        n, bps = int(wm.get("num_samples", 0) or 0), int(wm.get("avg_bytes_per_sample", 0) or 0)
        if n > 0 and bps > 0:
            bps_num += n * bps
            bps_den += n
    if not files:
        raise RuntimeError("Process self-play direct-save produced no chunk files.")
    merged = merge_self_play_stats(stats, max(1e-9, float(elapsed_sec)))
    meta = dict(metadata_base or {})
    meta.update({"self_play_target_samples_per_shard": int(target_samples_per_shard),
                 "self_play_chunk_target_bytes": int(chunk_target_bytes)})
    meta.update({k: merge_target_summaries(v) for k, v in summaries.items()})
    manifest = {"payload_format": "v1_sharded_manifest", "version": 1, "num_samples": int(sum(sizes)),
                "num_shards": len(files), "shard_files": files, "shard_sizes": sizes,
                "chunk_target_bytes": int(chunk_target_bytes), "avg_bytes_per_sample": int(bps_num // max(1, bps_den)),
                "stats": merged.to_dict(), "metadata": meta}
    os.makedirs(os.path.dirname(str(output_path)) or ".", exist_ok=True)
    torch.save(manifest, str(output_path))
    return manifest


def run_self_play_stage(*, model_state: Dict[str, torch.Tensor], num_games: int, devices: Sequence[str],
                        output_path: str, iteration_seed: int, mcts_simulations: int, temperature_init: float = 1.0,
                        temperature_final: float = 0.1, temperature_threshold: int = 10,
                        exploration_weight: float = 1.0, dirichlet_alpha: float = 0.3, dirichlet_epsilon: float = 0.25,
                        soft_value_k: float = 2.0, soft_label_alpha: float = 0.0, opening_random_moves: int = 0,
                        max_game_plies: int = 512, concurrent_games_per_device: int = 8192,
                        shard_dir: Optional[str] = None, target_samples_per_shard: int = 0, chunk_target_bytes: int = 0,
                        metadata_base: Optional[Dict[str, Any]] = None, sparse_ply: int = 1, sparse_top_k: int = 8,
                        search_backend: str = "cuda_root", portable_mcts_backend: str = "python",
                        portable_cpp_threads: int = 1, policy_target_temperature: Optional[float] = None,
                        policy_target_prior_pseudocount: float = 0.0, sample_moves: bool = True,
                        worker_fn: Optional[Callable[..., Dict[str, Any]]] = None,
                        in_process: bool = False) -> Tuple[SelfPlayV1Stats, Dict[str, Any]]:
    """Play `num_games` split over `devices` (one spawned process per device, each owning its GPU) and write
    `<stem>.wNN.chunkMMMMM<ext>` chunk files plus the manifest `output_path`.  Returns (merged stats, manifest).
    `worker_fn` / `in_process` exist for tests (a stub worker, no process pool)."""
    if worker_fn is None:
        from .self_play_worker import run_self_play_worker as worker_fn
    shards = split_games(int(num_games), len(devices))
    active = [(i, str(d), int(g)) for i, (d, g) in enumerate(zip(devices, shards)) if int(g) > 0]
    if not active:
        raise RuntimeError("No self-play shard assigned after game split.")
    own_workspace = not shard_dir
    workspace = shard_dir or tempfile.mkdtemp(prefix=f"lz_selfplay_{int(iteration_seed):06d}_")
    os.makedirs(workspace, exist_ok=True)
    state_path = os.path.join(workspace, "model_state_cpu.pt")
    torch.save({k: v.detach().cpu().clone() for k, v in model_state.items()}, state_path)
    out_dir = os.path.dirname(str(output_path)) or "."
    stem, ext = os.path.splitext(os.path.basename(str(output_path)))
    ext = ext or ".pt"
    os.makedirs(out_dir, exist_ok=True)

    def kwargs_for(idx: int, dev: str, games: int) -> Dict[str, Any]:
        return dict(
            worker_idx=idx, shard_device=dev, shard_games=games, seed=worker_seed(int(iteration_seed), idx),
            model_state_path=state_path,
            output_path=os.path.join(workspace, f"worker_manifest_{int(iteration_seed):06d}_{idx:02d}.pt"),
            mcts_simulations=int(mcts_simulations), temperature_init=float(temperature_init),
            temperature_final=float(temperature_final), temperature_threshold=int(temperature_threshold),
            exploration_weight=float(exploration_weight), dirichlet_alpha=float(dirichlet_alpha),
            dirichlet_epsilon=float(dirichlet_epsilon), soft_value_k=float(soft_value_k),
            opening_random_moves=int(opening_random_moves), max_game_plies=int(max_game_plies),
            concurrent_games_per_device=int(concurrent_games_per_device), soft_label_alpha=float(soft_label_alpha),
            sparse_ply=int(sparse_ply), sparse_top_k=int(sparse_top_k), search_backend=str(search_backend),
            portable_mcts_backend=str(portable_mcts_backend), portable_cpp_threads=int(portable_cpp_threads),
            policy_target_temperature=policy_target_temperature,
            policy_target_prior_pseudocount=float(policy_target_prior_pseudocount), sample_moves=bool(sample_moves),
            target_samples_per_shard=int(target_samples_per_shard), chunk_target_bytes=int(chunk_target_bytes),
            chunk_output_dir=out_dir, chunk_file_prefix=f"{stem}.w{idx:02d}", chunk_file_ext=ext)

    started = time.perf_counter()
    rows: List[Dict[str, Any]] = []
    failed = True
    try:
        # the workers share this node's cores and disk: each sizes its writer pool from its share (self_play_worker.
        # default_writer_threads); spawned children inherit the variable
        os.environ.setdefault("LZ_WORKERS_ON_NODE", str(max(1, len(active))))
        if in_process:
            rows = [worker_fn(**kwargs_for(*a)) for a in active]
        else:
            with ProcessPoolExecutor(max_workers=len(active), mp_context=mp.get_context("spawn")) as pool:
                futures = {pool.submit(worker_fn, **kwargs_for(*a)): a for a in active}
                for fut in as_completed(futures):
                    idx, dev, games = futures[fut]
                    try:
                        rows.append(fut.result())
                    except Exception as exc:
                        raise RuntimeError(f"self-play worker failed: worker={idx}, device={dev}, games={games}") from exc
        rows.sort(key=lambda r: int(r.get("worker_idx", 0)))
        manifest = merge_worker_manifests(
            [str(r["output_path"]) for r in rows], output_path=str(output_path),
            chunk_target_bytes=int(chunk_target_bytes), target_samples_per_shard=int(target_samples_per_shard),
            metadata_base=metadata_base, elapsed_sec=time.perf_counter() - started)
        failed = False
        return stats_from_payload(manifest["stats"]), manifest
    finally:
        if own_workspace and not failed:
            shutil.rmtree(workspace, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------
# writer / reader of whole payloads (trainer side)
# ---------------------------------------------------------------------------------------------------------
def save_sharded(*, path: str, samples: TensorSelfPlayBatch, stats_payload: Dict[str, Any], metadata: Dict[str, Any],
                 num_shards: int = 1, target_samples_per_shard: int = 0, chunk_target_bytes: int = 0) -> Dict[str, Any]:
    """Split an in-memory batch into chunk files `<stem>.shardNNNNN<ext>` + manifest (v1/train.py:1486-1559)."""
    out_dir = os.path.dirname(str(path)) or "."
    stem, ext = os.path.splitext(os.path.basename(str(path)))
    ext = ext or ".pt"
    bps = estimate_bytes_per_sample(samples)
    ranges = plan_sample_ranges(total_samples=samples.num_samples, num_shards=num_shards,
                                target_samples_per_shard=target_samples_per_shard,
                                chunk_target_bytes=chunk_target_bytes, bytes_per_sample=bps)
    files, sizes = [], []
    for i, (a, b) in enumerate(ranges):
        name = f"{stem}.shard{i:05d}{ext}"
        save_self_play_payload(path=os.path.join(out_dir, name), samples=slice_batch_cpu(samples, start=a, end=b),
                               stats_payload={}, metadata={"payload_format": "v1_sharded_shard", "shard_index": i,
                                                           "num_shards": len(ranges), "start": a, "end": b})
        files.append(name); sizes.append(b - a)
    cpu = samples.to("cpu")
    meta = dict(metadata)
    meta.setdefault("value_target_summary", summarize_scalar_targets(cpu.value_targets))
    meta.setdefault("soft_value_target_summary", summarize_scalar_targets(cpu.soft_value_targets))
    manifest = {"payload_format": "v1_sharded_manifest", "version": 1, "num_samples": int(samples.num_samples),
                "num_shards": len(files), "shard_files": files, "shard_sizes": sizes,
                "chunk_target_bytes": int(chunk_target_bytes), "avg_bytes_per_sample": int(bps),
                "stats": dict(stats_payload), "metadata": meta}
    os.makedirs(out_dir, exist_ok=True)
    torch.save(manifest, str(path))
    return manifest


def _batch_from_obj(obj: Any, where: str) -> TensorSelfPlayBatch:
    if isinstance(obj, TensorSelfPlayBatch):
        return obj.to("cpu")
    if not isinstance(obj, dict):
        raise RuntimeError(f"Unsupported shard format in {where}: {type(obj)!r}")
    missing = [k for k in TENSOR_KEYS if k not in obj]
    if missing:
        raise RuntimeError(f"Missing keys in shard {where}: {missing}")
    return TensorSelfPlayBatch(**{k: obj[k].to("cpu") for k in TENSOR_KEYS})


def _manifest_shards(manifest: Dict[str, Any], manifest_path: str) -> List[Tuple[str, int]]:
    base = os.path.dirname(manifest_path) or "."
    sizes = manifest.get("shard_sizes") or []
    out = []
    for i, entry in enumerate(manifest.get("shard_files") or []):
        name = str(entry).strip()
        if name:
            out.append((name if os.path.isabs(name) else os.path.join(base, name), int(sizes[i]) if i < len(sizes) else 0))
    return out


def _is_manifest(obj: Any) -> bool:
    return isinstance(obj, dict) and str(obj.get("payload_format", "")).strip().lower() == "v1_sharded_manifest"


def concat_batches(batches: Sequence[TensorSelfPlayBatch]) -> TensorSelfPlayBatch:
    if not batches:
        raise ValueError("no self-play batches to concatenate")
    return TensorSelfPlayBatch(**{k: torch.cat([getattr(b, k) for b in batches], dim=0) for k in TENSOR_KEYS})


def load_self_play_payload(path: str, *, ddp_rank: Optional[int] = None, ddp_world_size: Optional[int] = None
                           ) -> Tuple[TensorSelfPlayBatch, Dict[str, Any], Dict[str, Any]]:
    """Manifest or single payload -> (batch on CPU, stats, metadata); with a DDP rank/world the manifest's shards
    are dealt round-robin (v1/train.py:1610-1735)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Self-play payload not found: {path}")
    obj = _load(path)
    if not _is_manifest(obj):
        batch = _batch_from_obj(obj, path)
        get = (lambda k: dict(obj[k]) if isinstance(obj, dict) and isinstance(obj.get(k), dict) else {})
        return batch, get("stats"), get("metadata")
    shards = _manifest_shards(obj, path)
    if not shards:
        raise RuntimeError(f"Invalid sharded self-play manifest {path}: shard_files missing or empty.")
    picked = list(range(len(shards)))
    if ddp_rank is not None and ddp_world_size is not None and int(ddp_world_size) > 1:
        r, w = int(ddp_rank), int(ddp_world_size)
        if not 0 <= r < w:
            raise RuntimeError(f"Invalid ddp rank/world for shard load: rank={r}, world={w}")
        picked = [i for i in picked if i % w == r]
        if not picked:
            raise RuntimeError(f"DDP rank={r} got no shard from manifest={path} (world={w}, num_shards={len(shards)}).")
    merged = concat_batches([_batch_from_obj(_load(shards[i][0]), shards[i][0]) for i in picked])
    meta = dict(obj["metadata"]) if isinstance(obj.get("metadata"), dict) else {}
    meta.update({"payload_sharded_manifest": True, "payload_format": "v1_sharded_manifest", "manifest_path": str(path),
                 "manifest_num_shards": len(shards), "loaded_shard_indices": picked, "loaded_shard_count": len(picked),
                 "loaded_num_samples": int(merged.num_samples)})
    return merged, dict(obj["stats"]) if isinstance(obj.get("stats"), dict) else {}, meta


@dataclass
class ShardSpec:
    path: str
    num_samples: int
    sample_budget: int      # 0 = use every sample of the shard


def _spread_budget(sizes: Sequence[int], budget: int) -> List[int]:
    """Largest-remainder split of a replay sample budget over shards, never above a shard's size."""
    sizes = [max(0, int(s)) for s in sizes]
    total = sum(sizes)
    if not sizes or budget <= 0 or total <= 0:
        return [0] * len(sizes)
    target = min(int(budget), total)
    ideal = [target * s / total for s in sizes]
    alloc = [min(s, int(x)) for s, x in zip(sizes, ideal)]
    left = target - sum(alloc)
    for _, i in sorted(((-(ideal[i] - alloc[i]), i) for i in range(len(sizes)))):
        if left <= 0:
            break
        if alloc[i] < sizes[i]:
            alloc[i] += 1
            left -= 1
    return alloc


def resolve_shard_specs(primary_input: str, replay_inputs: Sequence[str], replay_budget_per_file: int, *,
                        ddp_rank: int = 0, ddp_world: int = 1) -> Tuple[List[ShardSpec], int]:
    """Shard paths + sample counts of the primary payload and the replay window, without loading tensors
    (streaming_dataset.py:69-140): replay files contribute at most `replay_budget_per_file` samples each."""
    specs: List[ShardSpec] = []
    total = 0
    for path, budget in [(primary_input, 0)] + [(r, int(replay_budget_per_file)) for r in replay_inputs]:
        if not os.path.exists(path):
            continue
        obj = _load(path)
        if _is_manifest(obj):
            shards = _manifest_shards(obj, path)
            if ddp_world > 1:
                shards = [s for i, s in enumerate(shards) if i % ddp_world == ddp_rank]
            budgets = _spread_budget([n for _, n in shards], budget) if budget > 0 else [0] * len(shards)
            for (full, n), b in zip(shards, budgets):
                specs.append(ShardSpec(full, n, b))
                total += min(n, b) if b > 0 else n
        else:
            n = int(obj.num_samples) if isinstance(obj, TensorSelfPlayBatch) else (
                int(obj["state_tensors"].shape[0]) if isinstance(obj, dict) and hasattr(obj.get("state_tensors"), "shape") else 0)
            specs.append(ShardSpec(path, n, budget))
            total += min(n, budget) if budget > 0 and n > 0 else n
    return specs, total


def iter_shard_batches(specs: Sequence[ShardSpec], *, seed: int = 0) -> Iterator[TensorSelfPlayBatch]:
    """Load one shard at a time (subsampled to its budget with a seeded permutation)."""
    g = torch.Generator().manual_seed(int(seed))
    for spec in specs:
        batch = _batch_from_obj(_load(spec.path), spec.path)
        if 0 < spec.sample_budget < batch.num_samples:
            keep = torch.randperm(batch.num_samples, generator=g)[: spec.sample_budget].sort().values
            batch = TensorSelfPlayBatch(**{k: getattr(batch, k).index_select(0, keep) for k in TENSOR_KEYS})
        yield batch
