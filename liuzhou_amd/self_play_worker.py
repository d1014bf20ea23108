"""Per-device self-play worker (mirror of v1/python/self_play_worker.py:276-552).

`run_self_play_worker(**kwargs)` takes the reference's 33 keyword arguments, runs the shard in chunks of
`concurrent_games_per_device` games, writes chunk payloads (`payload_format: v1_sharded_shard`) plus a worker
manifest (`payload_format: v1_worker_chunk_manifest`) and returns the same dict.  `search_backend`:
  "cuda_root" (reference default) -> root-PUCT over the HIP operators (self_play_v1_gpu)
  "portable" / "tree"            -> device-resident full-tree PUCT engine (self_play_tree_gpu)
Networks with 64 / 128 trunk channels run on the fused fp16-MFMA kernel; anything else uses PyTorch.
"""
from __future__ import annotations

import os
import time
import traceback
from typing import Any, Dict, List, Optional

import torch

from .net import ChessNet
from .self_play_storage import estimate_bytes_per_sample, plan_sample_ranges, save_self_play_payload, slice_batch_cpu
from .self_play_types import SelfPlayV1Stats

_SUMMARY_COUNTS = ("total", "finite_count", "nonfinite_count", "nonzero_count", "zero_count", "positive_count",
                   "negative_count", "near_zero_count", "ge_abs_0p05_count", "ge_abs_0p10_count", "ge_abs_0p20_count")


def _finish_summary(d: Dict[str, Any]) -> Dict[str, Any]:
    fin = max(1, int(d["finite_count"]))
    d["nonzero_ratio"] = float(d["nonzero_count"] / fin)
    d["abs_mean"] = float(d["sum_abs"] / fin)
    d["near_zero_ratio"] = float(d["near_zero_count"] / fin)
    for tag in ("0p05", "0p10", "0p20"):
        d[f"ge_abs_{tag}_ratio"] = float(d[f"ge_abs_{tag}_count"] / fin)
    return d


def summarize_scalar_targets(values: torch.Tensor) -> Dict[str, Any]:
    """Field set of self_play_worker.py:56-118."""
    d: Dict[str, Any] = {k: 0 for k in _SUMMARY_COUNTS}
    d["sum_abs"] = 0.0
    total = int(values.numel())
    if total > 0:
        finite = torch.isfinite(values)
        fv = values[finite]
        a = fv.abs()
        d.update(total=total, finite_count=int(fv.numel()), nonfinite_count=total - int(fv.numel()),
                 positive_count=int((fv > 0).sum()), negative_count=int((fv < 0).sum()), sum_abs=float(a.sum()),
                 near_zero_count=int((a <= 1e-6).sum()), ge_abs_0p05_count=int((a >= 0.05).sum()),
                 ge_abs_0p10_count=int((a >= 0.10).sum()), ge_abs_0p20_count=int((a >= 0.20).sum()))
        d["nonzero_count"] = d["positive_count"] + d["negative_count"]
        d["zero_count"] = d["finite_count"] - d["nonzero_count"]
    return _finish_summary(d)


def merge_target_summaries(summaries: List[Dict[str, Any]]) -> Dict[str, Any]:
    d: Dict[str, Any] = {k: 0 for k in _SUMMARY_COUNTS}
    d["sum_abs"] = 0.0
    for s in summaries:
        for k in _SUMMARY_COUNTS:
            d[k] += int(s.get(k, 0) or 0)
        d["sum_abs"] += float(s.get("sum_abs", 0.0) or 0.0)
    return _finish_summary(d)


def merge_self_play_stats(stats_list: List[SelfPlayV1Stats], elapsed_sec: float) -> SelfPlayV1Stats:
    keys = ("root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms")
    elapsed = max(1e-9, float(elapsed_sec))
    games = sum(int(s.num_games) for s in stats_list)
    positions = sum(int(s.num_positions) for s in stats_list)
    ms: Dict[str, float] = {k: 0.0 for k in keys}
    calls: Dict[str, int] = {k: 0 for k in keys}
    counters: Dict[str, int] = {}
    buckets = {str(d): 0 for d in range(-18, 19)}
    devices: List[str] = []
    for s in stats_list:
        for k, v in s.step_timing_ms.items():
            ms[k] = ms.get(k, 0.0) + float(v)
        for k, v in s.step_timing_calls.items():
            calls[k] = calls.get(k, 0) + int(v)
        for k, v in s.mcts_counters.items():
            counters[k] = counters.get(k, 0) + int(v)
        for k in buckets:
            buckets[k] += int((s.piece_delta_buckets or {}).get(k, 0) or 0)
        if s.device and s.device not in devices:
            devices.append(s.device)
    busy_ms = sum(max(0.0, float(s.elapsed_sec)) for s in stats_list) * 1000.0
    return SelfPlayV1Stats(
        num_games=games, num_positions=positions, black_wins=sum(int(s.black_wins) for s in stats_list),
        white_wins=sum(int(s.white_wins) for s in stats_list), draws=sum(int(s.draws) for s in stats_list),
        avg_game_length=float(sum(float(s.avg_game_length) * s.num_games for s in stats_list) / max(1, games)),
        elapsed_sec=elapsed, positions_per_sec=float(positions / elapsed), games_per_sec=float(games / elapsed),
        step_timing_ms=ms, step_timing_ratio={k: (min(1.0, max(0.0, v / busy_ms)) if busy_ms > 0 else 0.0) for k, v in ms.items()},
        step_timing_calls=calls, mcts_counters=counters, piece_delta_buckets=buckets, device=",".join(devices))


def _reserve_memory_anchor(device: torch.device) -> int:
    """V1_SELFPLAY_MEMORY_ANCHOR_MB (self_play_worker.py:33-53): keep a fixed allocation alive."""
    if device.type != "cuda":
        return 0
    try:
        mb = max(0, int(str(os.environ.get("V1_SELFPLAY_MEMORY_ANCHOR_MB", "")).strip() or 0))
    except ValueError:
        return 0
    if mb <= 0:
        return 0
    try:
        globals()["_MEMORY_ANCHOR"] = torch.empty((mb * 1024 * 1024,), dtype=torch.uint8, device=device)
    except Exception:
        return 0
    return mb


def _infer_model(state: Dict[str, torch.Tensor]) -> ChessNet:
    """Rebuild the architecture from the checkpoint's shapes (the reference always builds the 10x128 default;
    inferring keeps 6x64 checkpoints loadable too)."""
    trunk = int(state["stem_conv.weight"].shape[0])
    blocks = len({k.split(".")[1] for k in state if k.startswith("blocks.")})
    return ChessNet(trunk_channels=trunk, num_blocks=blocks,
                    policy_channels=int(state["policy_head.conv1.weight"].shape[0]),
                    value_channels=int(state["value_head.conv1.weight"].shape[0]),
                    value_mlp_channels=int(state["value_head.fc1.weight"].shape[0]),
                    value_bucket_bins=int(state["value_head.fc2.weight"].shape[0]))


def write_worker_chunks(run_once, *, worker_idx: int, device: str, games: int, games_per_chunk: int,
                        soft_label_alpha: float, chunk_dir: str, chunk_prefix: str, chunk_file_ext: str,
                        output_path: str, target_samples_per_shard: int, chunk_target_bytes: int,
                        meta_common: Dict[str, Any]) -> Dict[str, Any]:
    """Chunk loop of the worker (self_play_worker.py:430-546): `run_once(n) -> (TensorSelfPlayBatch, stats)` is called
    until `games` are played; every result is cut into `<prefix>.chunkNNNNN<ext>` payloads
    (`payload_format: v1_sharded_shard`) and the worker manifest (`v1_worker_chunk_manifest`) goes to `output_path`."""
    alpha = float(max(0.0, min(1.0, soft_label_alpha)))
    stats_chunks: List[SelfPlayV1Stats] = []
    val_s, soft_s, mix_s = [], [], []
    files: List[str] = []
    sizes: List[int] = []
    bps_num = bps_den = 0
    remaining = int(games)
    started = time.perf_counter()
    while remaining > 0:
        n = min(int(games_per_chunk), remaining)
        batch, st = run_once(n)
        if int((st.mcts_counters or {}).get("graph_retry_off", 0)):
            # a hipGraph capture failed in this chunk and the engine went on with direct launches (the reference records
            # the same about its finalize graph, v1/python/self_play_worker.py:434-442,475)
            meta_common["graph_retry_off"] = True
        cpu = batch.to("cpu")
        stats_chunks.append(st)
        val_s.append(summarize_scalar_targets(cpu.value_targets))
        soft_s.append(summarize_scalar_targets(cpu.soft_value_targets))
        mix_s.append(summarize_scalar_targets(torch.clamp((1.0 - alpha) * cpu.value_targets + alpha * cpu.soft_value_targets, -1.0, 1.0)))
        bps = estimate_bytes_per_sample(cpu)
        bps_num += bps * max(1, cpu.num_samples)
        bps_den += max(1, cpu.num_samples)
        for a, b in plan_sample_ranges(total_samples=cpu.num_samples, num_shards=1,
                                       target_samples_per_shard=int(target_samples_per_shard),
                                       chunk_target_bytes=int(chunk_target_bytes), bytes_per_sample=bps):
            name = f"{chunk_prefix}.chunk{len(files):05d}{chunk_file_ext}"
            meta = {"payload_format": "v1_sharded_shard", "worker_idx": int(worker_idx), "device": str(device),
                    "games": int(games), "games_per_chunk": int(games_per_chunk),
                    "num_selfplay_batches": len(stats_chunks), "saved_chunk_index": len(files)}
            meta.update(meta_common)
            meta["source_worker_manifest"] = os.path.basename(str(output_path))
            save_self_play_payload(path=os.path.join(chunk_dir, name), samples=slice_batch_cpu(cpu, start=a, end=b),
                                   stats_payload={}, metadata=meta)
            files.append(name)
            sizes.append(int(b - a))
        remaining -= n
    stats = merge_self_play_stats(stats_chunks, max(1e-9, time.perf_counter() - started))
    wmeta = {"worker_idx": int(worker_idx), "device": str(device), "games": int(games),
             "games_per_chunk": int(games_per_chunk), "num_selfplay_batches": len(stats_chunks),
             "saved_chunks": len(files)}
    wmeta.update(meta_common)
    manifest = {
        "payload_format": "v1_worker_chunk_manifest", "version": 1, "num_samples": int(sum(sizes)),
        "num_shards": len(files), "shard_files": list(files), "shard_sizes": list(sizes),
        "chunk_target_bytes": int(chunk_target_bytes), "avg_bytes_per_sample": int(bps_num // max(1, bps_den)),
        "stats": stats.to_dict(), "value_target_summary": merge_target_summaries(val_s),
        "soft_value_target_summary": merge_target_summaries(soft_s),
        "mixed_value_target_summary": merge_target_summaries(mix_s), "metadata": wmeta,
    }
    os.makedirs(os.path.dirname(str(output_path)) or ".", exist_ok=True)
    torch.save(manifest, str(output_path))
    return {"worker_idx": int(worker_idx), "device": str(device), "games": int(games), "output_path": str(output_path),
            "num_samples": int(sum(sizes)), "saved_chunks": len(files)}


def run_self_play_worker(*, worker_idx: int, shard_device: str, shard_games: int, seed: int, model_state_path: str,
                         output_path: str, mcts_simulations: int, temperature_init: float, temperature_final: float,
                         temperature_threshold: int, exploration_weight: float, dirichlet_alpha: float,
                         dirichlet_epsilon: float, soft_value_k: float, opening_random_moves: int,
                         max_game_plies: int, concurrent_games_per_device: int, soft_label_alpha: float = 0.0,
                         sample_moves: bool = True, target_samples_per_shard: int = 0, chunk_target_bytes: int = 0,
                         chunk_output_dir: Optional[str] = None, chunk_file_prefix: Optional[str] = None,
                         chunk_file_ext: str = ".pt", sparse_ply: int = 1, sparse_top_k: int = 8,
                         search_backend: str = "cuda_root", portable_mcts_backend: str = "python",
                         portable_cpp_threads: int = 1, policy_target_temperature: Optional[float] = None,
                         policy_target_prior_pseudocount: float = 0.0) -> Dict[str, Any]:
    try:
        torch.manual_seed(int(seed))
        dev = torch.device(str(shard_device))
        if dev.type != "cuda":
            raise RuntimeError("liuzhou_amd self-play worker needs a HIP device (no CPU path)")
        torch.cuda.set_device(dev)
        torch.cuda.manual_seed(int(seed))
        anchor_mb = _reserve_memory_anchor(dev)
        state = torch.load(str(model_state_path), map_location="cpu")
        if isinstance(state, dict) and "model_state_dict" in state:
            state = state["model_state_dict"]
        if not isinstance(state, dict):
            raise RuntimeError(f"Invalid model_state payload type: {type(state)!r} ({model_state_path})")
        games = int(shard_games)
        if games <= 0:
            raise ValueError(f"shard_games must be positive in worker, got {games}")
        concurrent = max(1, min(games, int(concurrent_games_per_device)))
        model = _infer_model(state)
        model.load_state_dict(state, strict=True)
        model.to(dev).eval()
        backend = str(search_backend).strip().lower()
        evaluator: Any = model
        if int(model.stem_conv.weight.shape[0]) in (64, 128):
            from .net_hip import FusedNet
            evaluator = FusedNet(model, dev)
        # other widths: the module itself is the (external fp32) evaluator of the tree engine / the root search
        chunk_dir, prefix = str(chunk_output_dir or "").strip(), str(chunk_file_prefix or "").strip()
        if not chunk_dir or not prefix:
            raise ValueError("run_self_play_worker requires chunk_output_dir and chunk_file_prefix to emit worker "
                             "manifest output.")

        chunk_no = [0]

        def run_once(n: int):
            chunk_no[0] += 1                  # every chunk plays NEW games: its own RNG key (game ids restart per chunk)
            rng_seed = (int(seed) * 1000003 + chunk_no[0]) & 0x7FFFFFFFFFFFFFFF
            common = dict(num_games=n, mcts_simulations=int(mcts_simulations), temperature_init=float(temperature_init),
                          temperature_final=float(temperature_final), temperature_threshold=int(temperature_threshold),
                          exploration_weight=float(exploration_weight), device=str(dev), add_dirichlet_noise=True,
                          dirichlet_alpha=float(dirichlet_alpha), dirichlet_epsilon=float(dirichlet_epsilon),
                          soft_value_k=float(soft_value_k), max_game_plies=int(max_game_plies),
                          sample_moves=bool(sample_moves), concurrent_games=max(1, min(n, concurrent)), verbose=False)
            if backend in ("portable", "tree"):
                from .tree_engine import self_play_tree_gpu
                return self_play_tree_gpu(evaluator, opening_random_moves=int(opening_random_moves),
                                          policy_target_temperature=policy_target_temperature,
                                          policy_target_prior_pseudocount=float(policy_target_prior_pseudocount),
                                          seed=rng_seed, collect_timing=True, **common)
            from .self_play_gpu_runner import self_play_v1_gpu
            return self_play_v1_gpu(evaluator, opening_random_moves=int(opening_random_moves), sparse_ply=int(sparse_ply),
                                    sparse_top_k=int(sparse_top_k), **common)

        meta_common = {"graph_retry_off": False, "memory_anchor_mb": int(anchor_mb),
                       "opening_random_moves": int(opening_random_moves), "search_backend": str(search_backend),
                       "portable_mcts_backend": str(portable_mcts_backend),
                       "portable_cpp_threads": int(portable_cpp_threads),
                       "policy_target_temperature": policy_target_temperature,
                       "policy_target_prior_pseudocount": float(policy_target_prior_pseudocount)}
        return write_worker_chunks(run_once, worker_idx=int(worker_idx), device=str(dev), games=games,
                                   games_per_chunk=concurrent, soft_label_alpha=float(soft_label_alpha),
                                   chunk_dir=chunk_dir, chunk_prefix=prefix, chunk_file_ext=str(chunk_file_ext),
                                   output_path=str(output_path), target_samples_per_shard=int(target_samples_per_shard),
                                   chunk_target_bytes=int(chunk_target_bytes), meta_common=meta_common)
    except Exception as exc:
        raise RuntimeError(f"v1 self-play process worker failed: worker={int(worker_idx)}, device={shard_device}, "
                           f"games={int(shard_games)}\n{traceback.format_exc()}") from exc
