"""Per-device self-play worker (mirror of v1/python/self_play_worker.py:276-552).

`run_self_play_worker(**kwargs)` takes the reference's 33 keyword arguments, plays the shard on
`concurrent_games_per_device` slots, writes chunk payloads (`payload_format: v1_sharded_shard`) plus a worker
manifest (`payload_format: v1_worker_chunk_manifest`) and returns the same dict.

Pipelined (round 5, `stream_worker_shard`): ONE wave plays the whole shard -- a finished slot starts the next game at
once, across what the reference calls chunks -- and the rows of finished games stream out while it plays: device
finished-row log (finished_log.py) -> pinned host staging on a side stream (copier thread) -> `torch.save` of the chunk
payloads (writer thread).  The reference's loop (`write_worker_chunks`: play a chunk to the end, `.to("cpu")`, save, with
the GPU idle; v1/python/self_play_worker.py:430-546) stays as the path for the options the device tail does not cover
and behind `LZ_WORKER_STREAM=0`.  Same file names, payload keys, dtypes and manifest keys either way.  `search_backend`:
  "cuda_root" (reference default) -> root-PUCT over the HIP operators (self_play_v1_gpu)
  "portable" / "tree"            -> device-resident full-tree PUCT engine (self_play_tree_gpu)
Networks with 64 / 128 trunk channels run on the fused fp16-MFMA kernel; anything else uses PyTorch.
"""
from __future__ import annotations

import os
import queue
import threading
import time
import traceback
from typing import Any, Callable, Dict, List, Optional

import torch

from .net import ChessNet
from .self_play_storage import estimate_bytes_per_sample, plan_sample_ranges, save_self_play_payload, slice_batch_cpu
from .self_play_types import SelfPlayV1Stats
from .trajectory_buffer import TensorSelfPlayBatch

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


def summarize_scalar_targets_np(values) -> Dict[str, Any]:
    """`summarize_scalar_targets` on a numpy view, single-threaded: the streaming writer runs beside the playing thread, and
    a torch CPU operator there wakes the whole intra-op thread pool (one thread per core of the host) under the playing
    thread's feet -- measured: the two halves of a C2 search stop overlapping and a run takes up to 1.5x as long."""
    import numpy as np
    d: Dict[str, Any] = {k: 0 for k in _SUMMARY_COUNTS}
    d["sum_abs"] = 0.0
    v = np.asarray(values).reshape(-1)
    total = int(v.size)
    if total > 0:
        fv = v[np.isfinite(v)]
        a = np.abs(fv)
        d.update(total=total, finite_count=int(fv.size), nonfinite_count=total - int(fv.size),
                 positive_count=int((fv > 0).sum()), negative_count=int((fv < 0).sum()),
                 sum_abs=float(a.sum(dtype=np.float32)), near_zero_count=int((a <= 1e-6).sum()),
                 ge_abs_0p05_count=int((a >= 0.05).sum()), ge_abs_0p10_count=int((a >= 0.10).sum()),
                 ge_abs_0p20_count=int((a >= 0.20).sum()))
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


def workers_on_node() -> int:
    """Self-play workers that share this node's cores, page cache and disk: LZ_WORKERS_ON_NODE (set by
    `run_self_play_stage` for the processes it spawns), else torchrun's LOCAL_WORLD_SIZE / WORLD_SIZE, else 1."""
    for key in ("LZ_WORKERS_ON_NODE", "LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = str(os.environ.get(key, "")).strip()
        if v.isdigit() and int(v) > 0:
            return int(v)
    return 1


def default_writer_threads() -> int:
    """Writer threads of one streaming worker when LZ_WORKER_WRITERS is not set: three (what one R worker's drain burst
    needs, round 5) while the node has the cores -- each worker also runs its playing thread and its copier, so a worker's
    share of the cores this process may use, minus those two, caps it: 8 workers on a 128-core node keep 3 writers each
    (40 host threads), 8 workers on 16 cores get 1 (24 threads) instead of oversubscribing the cores 2.5x.  Measured by
    scripts/rehearse_host_io.py (profiles/r06_host_io_rehearsal.md)."""
    env = str(os.environ.get("LZ_WORKER_WRITERS", "")).strip()
    if env.isdigit() and int(env) > 0:
        return int(env)
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(1, min(3, cores // workers_on_node() - 2))


def configure_save(crc32: Optional[bool] = None) -> bool:
    """`torch.save` computes a crc32 of every record: about half of a writer thread's time on a payload that is one big
    memcpy otherwise.  LZ_SAVE_CRC32=0 (or crc32=False) turns it off for this process -- `torch.load`, and so the
    reference's loader, reads such files unchanged; external unzip tools may warn.  Default: on (the reference's files
    carry checksums).  Returns the setting in force."""
    if crc32 is None:
        crc32 = os.environ.get("LZ_SAVE_CRC32", "1") != "0"
    try:
        torch.serialization.set_crc32_options(bool(crc32))
        return bool(torch.serialization.get_crc32_options())
    except AttributeError:                                      # older torch: always on
        return True


class ShardStreamer:
    """Host half of the streaming worker: takes the segments the finished-row log cuts (`on_segment`, called on the playing
    thread: a queue put), copies each to pinned staging on its own stream (copier thread; the log arena goes back to the
    player as soon as the copy has landed) and writes the chunk payloads (writer thread; `torch.save` releases the GIL).
    `plan_segment(rows, number)` runs on the copier thread, in segment order (file names are assigned there);
    `write_segment(staging, rows, plan)` on one of `writers` threads (a `torch.save` is one core's memcpy + crc32, 2.6 - 4.4
    GB/s on the GPU box: three writers keep up with the burst of games that end while the last wave drains).  One staging
    buffer more than writers: the copy of the next segment overlaps the writes; writers that cannot keep up hold the staging
    buffers, then the log arenas, then -- through the kernels' back-pressure -- the slots of finished games."""

    def __init__(self, device: torch.device, capacity_rows: int, action_dim: int, plan_segment: Callable[..., Any],
                 write_segment: Callable[..., None], writers: int = 3) -> None:
        self.device, self.capacity, self.A = torch.device(device), int(capacity_rows), int(action_dim)
        self.plan_segment, self.write_segment = plan_segment, write_segment
        self.n_writers = max(1, int(writers))
        self.n_staging = self.n_writers + 1
        self.segments: "queue.Queue" = queue.Queue()
        self.copied: "queue.Queue" = queue.Queue()
        self.free_staging: "queue.Queue" = queue.Queue()
        self.error: Optional[BaseException] = None
        self._closed = False
        self.rows = self.games = 0
        self.copy_ms = self.write_ms = 0.0
        self._lock = threading.Lock()
        self._threads = [threading.Thread(target=self._guard, args=(self._copier,), name="lz-shard-copier", daemon=True)] + [
            threading.Thread(target=self._guard, args=(self._writer,), name=f"lz-shard-writer{i}", daemon=True)
            for i in range(self.n_writers)]
        for t in self._threads:
            t.start()

    def on_segment(self, seg) -> None:
        self.raise_if_failed()
        self.segments.put(seg)

    def raise_if_failed(self) -> None:
        if self.error is not None:
            raise RuntimeError("streaming shard writer failed") from self.error

    def _guard(self, fn) -> None:
        try:
            fn()
        except BaseException as exc:  # noqa: BLE001 -- handed to the playing thread (on_segment / on_blocked / finish)
            if self.error is None:
                self.error = exc

    def _get(self, q: "queue.Queue"):
        """Blocking get that gives up when the other thread (or the player) has failed."""
        while True:
            try:
                return q.get(timeout=0.1)
            except queue.Empty:
                if self.error is not None or self._closed:
                    raise RuntimeError("streaming shard writer: stopped") from self.error

    def _copier(self) -> None:
        torch.cuda.set_device(self.device)
        side = torch.cuda.Stream(self.device)
        cap, A = self.capacity, self.A
        allocated = 0
        while True:
            seg = self._get(self.segments)
            # Staging buffers are pinned on demand, one tensor at a time: pinning blocks the other threads' HIP calls
            # while it runs (~80 ms per GB measured), and the playing thread is only two plies ahead of the device
            try:
                staging = self.free_staging.get_nowait()
            except queue.Empty:
                if allocated < self.n_staging:
                    staging = (torch.empty((cap, 11, 6, 6), dtype=torch.float32, pin_memory=True),
                               torch.empty((cap, A), dtype=torch.bool, pin_memory=True),
                               torch.empty((cap, A), dtype=torch.float32, pin_memory=True),
                               torch.empty((cap,), dtype=torch.float32, pin_memory=True),
                               torch.empty((cap,), dtype=torch.float32, pin_memory=True))
                    allocated += 1
                else:
                    staging = self._get(self.free_staging)
            t0 = time.perf_counter()
            with torch.cuda.stream(side):
                side.wait_event(seg.ready)
                rows, games = (int(x) for x in seg.arena.counters[:2].tolist())     # waits for `ready` only (side stream)
                a = seg.arena
                for dst, src in zip(staging, (a.state, a.legal, a.policy, a.value, a.soft)):
                    dst[:rows].copy_(src[:rows], non_blocking=True)
                side.synchronize()
            seg.release()                                        # the player may fill this arena again
            self.copy_ms += (time.perf_counter() - t0) * 1e3
            self.copied.put((staging, rows, games, self.plan_segment(rows, seg.number) if rows > 0 else None))
            if seg.final:
                break
        for _ in range(self.n_writers):
            self.copied.put(None)

    def _writer(self) -> None:
        while True:
            item = self._get(self.copied)
            if item is None:
                break
            staging, rows, games, plan = item
            t0 = time.perf_counter()
            if rows > 0 and os.environ.get("LZ_WORKER_NOWRITE", "0") != "1":     # experiment switch: copy out, write nothing
                self.write_segment(staging, rows, plan)
            with self._lock:
                self.rows += rows
                self.games += games
                self.write_ms += (time.perf_counter() - t0) * 1e3              # summed over the writers
            self.free_staging.put(staging)

    def finish(self) -> None:
        """Wait until every segment is on disk (call after the runner has closed the log)."""
        for t in self._threads:
            while t.is_alive() and self.error is None:
                t.join(timeout=0.2)
        self.abort()
        self.raise_if_failed()

    def abort(self) -> None:
        """Stop the threads (they notice within 0.1 s) -- the player has failed, or everything is written."""
        self._closed = True


def _ROW_BYTES(action_dim: int) -> int:
    """Bytes of one sample in the five payload tensors (estimate_bytes_per_sample of a full batch): 2 692 for 220 actions."""
    return 11 * 36 * 4 + int(action_dim) * (1 + 4) + 4 + 4


def _host_view(t: torch.Tensor, start: int, end: int) -> torch.Tensor:
    """Rows [start, end) of a host tensor as a tensor whose storage is exactly those bytes (no copy): `torch.save` writes a
    tensor's whole storage, and the staging buffers are as large as a log arena."""
    return torch.from_numpy(t.numpy()[int(start):int(end)])


class StreamedShardFiles:
    """Host-only half of the streamed shard: which rows go to which payload file (`plan_segment`, called in segment order),
    the `torch.save` of a segment's files from host staging tensors (`write_segment`, any thread) and the worker manifest
    (`finish`).  Same file names, payload keys and manifest keys as the reference's chunk loop
    (v1/python/self_play_worker.py:430-546); a file holds the games that ended within one segment of the finished-row
    log, cut further by `chunk_target_bytes` / `target_samples_per_shard` exactly as the reference cuts a chunk
    (`plan_sample_ranges`).  No device in here: tests/test_self_play_stage.py drives it on the CPU and reads the result
    back through the reference's own loader."""

    def __init__(self, *, device, worker_idx: int, games: int, games_per_chunk: int, soft_label_alpha: float,
                 chunk_dir: str, chunk_prefix: str, chunk_file_ext: str, output_path: str, target_samples_per_shard: int,
                 chunk_target_bytes: int, meta_common: Dict[str, Any], action_dim: int = 220) -> None:
        self.device, self.worker_idx, self.games, self.games_per_chunk = str(device), int(worker_idx), int(games), int(games_per_chunk)
        self.alpha = float(max(0.0, min(1.0, soft_label_alpha)))
        self.chunk_dir, self.prefix, self.ext, self.output_path = str(chunk_dir), str(chunk_prefix), str(chunk_file_ext), str(output_path)
        self.target_samples, self.target_bytes = int(target_samples_per_shard), int(chunk_target_bytes)
        self.meta_common, self.action_dim = meta_common, int(action_dim)
        self.val_s: List[Dict[str, Any]] = []
        self.soft_s: List[Dict[str, Any]] = []
        self.mix_s: List[Dict[str, Any]] = []
        self.files: List[str] = []
        self.sizes: List[int] = []
        self.bps = [0, 0]
        self.lock = threading.Lock()
        os.makedirs(self.chunk_dir or ".", exist_ok=True)

    def plan_segment(self, rows: int, number: int):
        """File names and row ranges of one segment (one thread, segment order)."""
        plan = []
        for lo, hi in plan_sample_ranges(total_samples=rows, num_shards=1, target_samples_per_shard=self.target_samples,
                                         chunk_target_bytes=self.target_bytes, bytes_per_sample=_ROW_BYTES(self.action_dim)):
            name = f"{self.prefix}.chunk{len(self.files):05d}{self.ext}"
            meta = {"payload_format": "v1_sharded_shard", "worker_idx": self.worker_idx, "device": self.device,
                    "games": self.games, "games_per_chunk": self.games_per_chunk,
                    "num_selfplay_batches": int(number) + 1, "saved_chunk_index": len(self.files)}
            meta.update(self.meta_common)
            meta["source_worker_manifest"] = os.path.basename(self.output_path)
            plan.append((name, lo, hi, meta))
            self.files.append(name)
            self.sizes.append(int(hi - lo))
        return plan

    def write_segment(self, staging, rows: int, plan) -> None:
        import numpy as np
        state, legal, policy, value, soft = (t[:rows] for t in staging)
        v_np, s_np = value.numpy(), soft.numpy()                  # numpy, not torch: no intra-op thread pool (see above)
        a = self.alpha
        summaries = (summarize_scalar_targets_np(v_np), summarize_scalar_targets_np(s_np),
                     summarize_scalar_targets_np(np.clip(np.float32(1.0 - a) * v_np + np.float32(a) * s_np,
                                                         np.float32(-1.0), np.float32(1.0))))
        b = estimate_bytes_per_sample(TensorSelfPlayBatch(state, legal, policy, value, soft))
        with self.lock:
            self.val_s.append(summaries[0]); self.soft_s.append(summaries[1]); self.mix_s.append(summaries[2])
            self.bps[0] += b * max(1, rows)
            self.bps[1] += max(1, rows)
        for name, lo, hi, meta in plan:
            save_self_play_payload(path=os.path.join(self.chunk_dir, name),
                                   samples=TensorSelfPlayBatch(*(_host_view(t, lo, hi) for t in staging)),
                                   stats_payload={}, metadata=meta)

    def finish(self, stats: SelfPlayV1Stats, num_batches: int) -> Dict[str, Any]:
        """Write the worker manifest (`v1_worker_chunk_manifest`) and return the worker's result row."""
        wmeta = {"worker_idx": self.worker_idx, "device": self.device, "games": self.games,
                 "games_per_chunk": self.games_per_chunk, "num_selfplay_batches": int(num_batches),
                 "saved_chunks": len(self.files)}
        wmeta.update(self.meta_common)
        manifest = {
            "payload_format": "v1_worker_chunk_manifest", "version": 1, "num_samples": int(sum(self.sizes)),
            "num_shards": len(self.files), "shard_files": list(self.files), "shard_sizes": list(self.sizes),
            "chunk_target_bytes": self.target_bytes, "avg_bytes_per_sample": int(self.bps[0] // max(1, self.bps[1])),
            "stats": stats.to_dict(), "value_target_summary": merge_target_summaries(self.val_s),
            "soft_value_target_summary": merge_target_summaries(self.soft_s),
            "mixed_value_target_summary": merge_target_summaries(self.mix_s), "metadata": wmeta,
        }
        os.makedirs(os.path.dirname(self.output_path) or ".", exist_ok=True)
        torch.save(manifest, self.output_path)
        return {"worker_idx": self.worker_idx, "device": self.device, "games": self.games, "output_path": self.output_path,
                "num_samples": int(sum(self.sizes)), "saved_chunks": len(self.files)}


def stream_worker_shard(play, *, device: torch.device, worker_idx: int, games: int, games_per_chunk: int,
                        max_game_plies: int, soft_label_alpha: float, chunk_dir: str, chunk_prefix: str,
                        chunk_file_ext: str, output_path: str, target_samples_per_shard: int, chunk_target_bytes: int,
                        meta_common: Dict[str, Any], segment_games: Optional[int] = None,
                        action_dim: int = 220, extra_counters: Optional[Dict[str, int]] = None) -> Dict[str, Any]:
    """The pipelined worker: `play(row_log) -> SelfPlayV1Stats` runs the whole shard on `games_per_chunk` slots with the
    finished-row log attached; `segment_games` (default an eighth of `games_per_chunk`, so that what is left to write when
    the last game ends is small; env LZ_WORKER_SEGMENT_GAMES) finished games make a segment = one payload file unless
    `chunk_target_bytes` / `target_samples_per_shard` cut it further (`StreamedShardFiles`)."""
    from .finished_log import FinishedRowLog
    if segment_games is None:
        env = str(os.environ.get("LZ_WORKER_SEGMENT_GAMES", "")).strip()
        segment_games = int(env) if env else max(1, int(games_per_chunk) // 8)
    shard = StreamedShardFiles(device=device, worker_idx=worker_idx, games=games, games_per_chunk=games_per_chunk,
                               soft_label_alpha=soft_label_alpha, chunk_dir=chunk_dir, chunk_prefix=chunk_prefix,
                               chunk_file_ext=chunk_file_ext, output_path=output_path,
                               target_samples_per_shard=target_samples_per_shard, chunk_target_bytes=chunk_target_bytes,
                               meta_common=meta_common, action_dim=action_dim)
    t_stream = time.perf_counter()
    wave = max(1, min(int(games), int(games_per_chunk)))
    env_rows = str(os.environ.get("LZ_WORKER_LOG_ROWS", "")).strip()      # rows per log arena (tests: force back-pressure)
    log = FinishedRowLog(device, segment_games=int(segment_games), num_slots=wave, max_steps=int(max_game_plies),
                         action_dim=int(action_dim), capacity_rows=int(env_rows) if env_rows else None)
    configure_save()
    streamer = ShardStreamer(device, log.capacity, int(action_dim), shard.plan_segment, shard.write_segment,
                             writers=default_writer_threads())
    log.on_segment = streamer.on_segment
    log.on_blocked = streamer.raise_if_failed
    started = time.perf_counter()
    log_setup_ms = int((started - t_stream) * 1e3)
    try:
        st = play(log)
    except BaseException:
        streamer.abort()
        raise
    play_sec = time.perf_counter() - started
    streamer.finish()
    if int((st.mcts_counters or {}).get("graph_retry_off", 0)):
        meta_common["graph_retry_off"] = True
    if streamer.rows != int(st.num_positions):
        raise RuntimeError(f"streaming worker: {streamer.rows} rows written, the runner recorded {int(st.num_positions)}")
    stats = merge_self_play_stats([st], max(1e-9, time.perf_counter() - started))
    stats.mcts_counters.update({"stream_segments": int(log.segments_cut), "stream_blocked_polls": int(log.blocked_polls),
                                "stream_copy_ms": int(streamer.copy_ms), "stream_write_ms": int(streamer.write_ms),
                                "stream_tail_ms": int((time.perf_counter() - started - play_sec) * 1e3),
                                "stream_setup_ms": log_setup_ms, "play_call_ms": int(play_sec * 1e3),
                                **(extra_counters or {})})
    return shard.finish(stats, int(log.segments_cut))


def streaming_footprint(num_slots: int, max_game_plies: int, *, segment_games: Optional[int] = None, action_dim: int = 220,
                        writers: int = 3, log_rows: Optional[int] = None) -> Dict[str, int]:
    """Bytes the pipelined worker (`stream_worker_shard`) allocates, whatever the game lengths turn out to be: the
    slot-major live arena holds `num_slots x max_game_plies` rows (the chunk loop's cursor arena only the ~130 rows a game
    really has), two log arenas on the device and `writers + 1` pinned staging buffers of one log arena each on the
    host.  16 384 slots x 512 plies: 22.6 GB + 2 x 1.0 GB on the device, 4 x 1.0 GB pinned."""
    row = _ROW_BYTES(int(action_dim)) + 1                      # + the player-sign byte of the live arena
    seg = max(1, int(num_slots) // 8) if segment_games is None else max(1, int(segment_games))
    cap = int(log_rows) if log_rows else max(seg * 160 + 2 * int(num_slots), 2 * int(max_game_plies))
    return {"live_arena_bytes": int(num_slots) * int(max_game_plies) * row, "log_arena_bytes": 2 * cap * row,
            "pinned_host_bytes": (int(writers) + 1) * cap * row,
            "device_bytes": int(num_slots) * int(max_game_plies) * row + 2 * cap * row}


def streaming_fits(dev: torch.device, num_slots: int, max_game_plies: int, share: float = 0.5, **kw) -> Optional[str]:
    """None when the streaming worker's device footprint is at most `share` of the memory the device can still give
    (the tree engine sizes its arenas from what is left afterwards), else the reason to fall back to the chunk loop --
    which keeps ~130 rows per game instead of `max_game_plies` (a shard that fitted before round 5 must not
    fail inside its first ply).  LZ_WORKER_STREAM_SHARE overrides `share`."""
    from .distributed import _free_device_bytes
    share = float(os.environ.get("LZ_WORKER_STREAM_SHARE", share))
    need = streaming_footprint(num_slots, max_game_plies, **kw)["device_bytes"]
    free = _free_device_bytes(dev)
    if free >= 0 and need > share * free:
        return (f"streaming worker needs {need / 2**30:.1f} GiB on the device ({num_slots} slots x {max_game_plies} plies, "
                f"slot-major) and {share:.0%} of the free {free / 2**30:.1f} GiB is less: chunk loop instead")
    return None


def pick_evaluator(model, dev: torch.device):
    """(evaluator, "fused_f16" | "torch", reason or None) for a worker / stage: the fused kernel when it is built for the
    checkpoint's shape (`net_hip.fused_supported`: trunk width, block count AND head sizes -- the reference's ChessNet is
    generic in all of them, src/neural_network.py:213-246), else the module itself on `dev`, as the reference's worker
    evaluates (v1/python/self_play_worker.py:335-338).  Never raises for a shape: a worker must not die, or silently
    change what it computes, on a checkpoint the trainer was able to produce."""
    from .net_hip import FusedNet, fused_unsupported_reason
    why = fused_unsupported_reason(model)
    if why is None:
        return FusedNet(model, dev), "fused_f16", None       # packed from the host copy: the module never visits the device
    model.to(dev)             # the module is the (external fp32) evaluator of the tree engine / the root search's `model`
    return model, "torch", why


def _drop_engines() -> None:
    """The worker's network dies with the call, so the tree engines cached on it (tens of GB of arenas) go too."""
    try:
        from .tree_engine import clear_engine_cache
        clear_engine_cache()
    except Exception:
        pass


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
        t_worker = time.perf_counter()
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
        model.eval()
        backend = str(search_backend).strip().lower()
        evaluator, evaluator_name, evaluator_why = pick_evaluator(model, dev)
        chunk_dir, prefix = str(chunk_output_dir or "").strip(), str(chunk_file_prefix or "").strip()
        if not chunk_dir or not prefix:
            raise ValueError("run_self_play_worker requires chunk_output_dir and chunk_file_prefix to emit worker "
                             "manifest output.")

        chunk_no = [0]
        tree = backend in ("portable", "tree")
        stream = os.environ.get("LZ_WORKER_STREAM", "1") != "0"
        if stream and not tree:
            from .self_play_gpu_runner import streaming_supported
            stream = streaming_supported(evaluator, opening_random_moves=int(opening_random_moves), sparse_ply=int(sparse_ply))
            # (every reference option of the R backend -- sparse_ply, child_eval_mode, opening moves -- now stays on the fused,
            #  sync-free search: streaming is only off for a network the fused kernel is not built for)

        stream_fallback = None
        if stream:
            env = os.environ
            stream_fallback = streaming_fits(
                dev, concurrent, int(max_game_plies), writers=default_writer_threads(),
                segment_games=int(env["LZ_WORKER_SEGMENT_GAMES"]) if env.get("LZ_WORKER_SEGMENT_GAMES", "").strip() else None,
                log_rows=int(env["LZ_WORKER_LOG_ROWS"]) if env.get("LZ_WORKER_LOG_ROWS", "").strip() else None)
            if stream_fallback is not None:
                print(f"[liuzhou_amd] worker {int(worker_idx)}: {stream_fallback}", flush=True)
                stream = False

        def run_once(n: int, row_log=None):
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
                                          seed=rng_seed, collect_timing=os.environ.get("LZ_WORKER_TIMING", "1") != "0", row_log=row_log,
                                          **common)
            from .self_play_gpu_runner import self_play_v1_gpu
            return self_play_v1_gpu(evaluator, opening_random_moves=int(opening_random_moves), sparse_ply=int(sparse_ply),
                                    sparse_top_k=int(sparse_top_k), row_log=row_log, **common)

        setup_ms = int((time.perf_counter() - t_worker) * 1e3)     # checkpoint load, module, network packing
        meta_common = {"graph_retry_off": False, "memory_anchor_mb": int(anchor_mb),
                       "opening_random_moves": int(opening_random_moves), "search_backend": str(search_backend),
                       "portable_mcts_backend": str(portable_mcts_backend),
                       "portable_cpp_threads": int(portable_cpp_threads),
                       "policy_target_temperature": policy_target_temperature,
                       "policy_target_prior_pseudocount": float(policy_target_prior_pseudocount),
                       # which network evaluator played this shard: the hand-written kernel or the module through
                       # PyTorch (a shape the kernel is not built for; `evaluator_reason` says which)
                       "evaluator": evaluator_name, **({"evaluator_reason": evaluator_why} if evaluator_why else {}),
                       "streamed": bool(stream), **({"stream_fallback": stream_fallback} if stream_fallback else {})}
        if stream:
            os.makedirs(chunk_dir, exist_ok=True)
            return stream_worker_shard(lambda log: run_once(games, row_log=log)[1], device=dev, worker_idx=int(worker_idx),
                                       games=games, games_per_chunk=concurrent, max_game_plies=int(max_game_plies),
                                       soft_label_alpha=float(soft_label_alpha), chunk_dir=chunk_dir, chunk_prefix=prefix,
                                       chunk_file_ext=str(chunk_file_ext), output_path=str(output_path),
                                       target_samples_per_shard=int(target_samples_per_shard),
                                       chunk_target_bytes=int(chunk_target_bytes), meta_common=meta_common,
                                       extra_counters={"worker_setup_ms": setup_ms})
        return write_worker_chunks(run_once, worker_idx=int(worker_idx), device=str(dev), games=games,
                                   games_per_chunk=concurrent, soft_label_alpha=float(soft_label_alpha),
                                   chunk_dir=chunk_dir, chunk_prefix=prefix, chunk_file_ext=str(chunk_file_ext),
                                   output_path=str(output_path), target_samples_per_shard=int(target_samples_per_shard),
                                   chunk_target_bytes=int(chunk_target_bytes), meta_common=meta_common)
    except Exception as exc:
        raise RuntimeError(f"v1 self-play process worker failed: worker={int(worker_idx)}, device={shard_device}, "
                           f"games={int(shard_games)}\n{traceback.format_exc()}") from exc
    finally:
        # the per-call FusedNet is the cache key of the tree engines built on it: it can never hit again, and its arenas
        # (several GB at C2, tens of GB at C3) would outlive the worker next to an in-process trainer
        _drop_engines()
