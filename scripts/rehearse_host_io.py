"""Rehearsal of the HOST side of N self-play workers on one node, without GPUs.

Eight workers of C4 share one node's cores, page cache and disk.  Each worker's host side is `ShardStreamer`'s file half:
a copier hands segments (staging buffers of one log arena) to `writers` threads that `torch.save` them through
`StreamedShardFiles` (v1/python/self_play_worker.py:464-537 file format).  This program starts `--procs` processes; each one
produces synthetic segments at `--rate-gbps` (the R worker's measured 1.4 GB/s: 520 k positions/s x 2 692 B) into ONE
directory, with the real `StreamedShardFiles.plan_segment / write_segment / finish`, `writers + 1` staging buffers and the
same back-pressure rule (the producer stalls when no staging buffer is free -- on the GPU that is what ends up holding
finished games in their slots).  Reported per process: GB written, GB/s, seconds the producer was stalled (= the time the
GPU side would have been back-pressured), staging bytes; and the aggregate.  `--writers 0` = the default rule of the worker
(`default_writer_threads`: from the cores this process may use and the workers on the node).

usage: python scripts/rehearse_host_io.py --procs 8 --gb-per-proc 2 --rate-gbps 1.4 [--writers 0] [--dir /tmp/x] [--crc32 0]
Prints one JSON line (also the CPU test's interface: tests/test_host_io_rehearsal.py runs it small)."""
import argparse
import json
import multiprocessing as mp
import os
import queue
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank: int, args, out_q) -> None:
    os.environ["LZ_WORKERS_ON_NODE"] = str(args.procs)
    import torch
    torch.set_num_threads(1)
    from liuzhou_amd.self_play_worker import StreamedShardFiles, default_writer_threads, _ROW_BYTES, configure_save
    from liuzhou_amd.self_play_types import SelfPlayV1Stats
    configure_save(crc32=bool(args.crc32))
    writers = int(args.writers) if int(args.writers) > 0 else default_writer_threads()
    row = _ROW_BYTES(220)
    seg_rows = max(1, int(args.segment_mb * (1 << 20)) // row)
    total_rows = max(seg_rows, int(args.gb_per_proc * (1 << 30)) // row)
    shard = StreamedShardFiles(device=f"cuda:{rank}", worker_idx=rank, games=total_rows // 130 + 1, games_per_chunk=8192,
                               soft_label_alpha=0.0, chunk_dir=args.dir, chunk_prefix=f"it000.w{rank:02d}", chunk_file_ext=".pt",
                               output_path=os.path.join(args.dir, f"worker_{rank:02d}.pt"), target_samples_per_shard=0,
                               chunk_target_bytes=int(args.chunk_target_mb * (1 << 20)), meta_common={"rehearsal": True})
    # staging buffers: writers + 1, as ShardStreamer keeps them (plain host memory here: no device to pin for)
    g = torch.Generator().manual_seed(rank)

    def staging():
        pol = torch.rand((seg_rows, 220), generator=g)
        return (torch.rand((seg_rows, 11, 6, 6), generator=g).round_(), pol > 0.7, pol / pol.sum(1, keepdim=True),
                torch.rand((seg_rows,), generator=g) * 2 - 1, torch.rand((seg_rows,), generator=g) * 2 - 1)
    free: "queue.Queue" = queue.Queue()
    for _ in range(writers + 1):
        free.put(staging())
    todo: "queue.Queue" = queue.Queue()
    write_busy = [0.0]
    lock = threading.Lock()

    def writer():
        while True:
            item = todo.get()
            if item is None:
                return
            st, rows, plan = item
            t0 = time.perf_counter()
            shard.write_segment(st, rows, plan)
            with lock:
                write_busy[0] += time.perf_counter() - t0
            free.put(st)
    threads = [threading.Thread(target=writer, daemon=True) for _ in range(writers)]
    for t in threads:
        t.start()
    out_q.put(("ready", rank))
    args.start_evt.wait()
    t_start = time.perf_counter()
    produced, number, stalled = 0, 0, 0.0
    while produced < total_rows:
        rows = min(seg_rows, total_rows - produced)
        # the GPU side delivers a segment every rows * row_bytes / rate seconds: wait for that instant, then for a buffer
        due = t_start + (produced + rows) * row / (args.rate_gbps * 1e9)
        now = time.perf_counter()
        if due > now:
            time.sleep(due - now)
        t0 = time.perf_counter()
        st = free.get()
        stalled += time.perf_counter() - t0                       # back-pressure: no staging buffer was free
        todo.put((st, rows, shard.plan_segment(rows, number)))
        produced += rows
        number += 1
    for _ in threads:
        todo.put(None)
    for t in threads:
        t.join()
    wall = time.perf_counter() - t_start
    stats = SelfPlayV1Stats(num_games=total_rows // 130 + 1, num_positions=total_rows, black_wins=0, white_wins=0, draws=0,
                            avg_game_length=130.0, elapsed_sec=wall, positions_per_sec=total_rows / wall,
                            games_per_sec=0.0, step_timing_ms={}, step_timing_ratio={}, step_timing_calls={},
                            mcts_counters={}, piece_delta_buckets={}, device=f"cuda:{rank}")
    res = shard.finish(stats, number)
    out_q.put(("done", {"rank": rank, "gb": total_rows * row / 1e9, "wall_s": round(wall, 3),
                        "gbps": round(total_rows * row / 1e9 / wall, 3), "producer_stalled_s": round(stalled, 3),
                        "writer_busy_s": round(write_busy[0], 3), "writers": writers, "files": int(res["saved_chunks"]),
                        "staging_bytes": (writers + 1) * seg_rows * (row + 0), "rows": int(res["num_samples"])}))


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--gb-per-proc", type=float, default=2.0)
    ap.add_argument("--rate-gbps", type=float, default=1.4, help="per process: what its GPU delivers (R worker: 1.4)")
    ap.add_argument("--writers", type=int, default=0, help="writer threads per process (0 = the worker's default rule)")
    ap.add_argument("--segment-mb", type=float, default=336.0, help="one segment = 1 024 games x ~125 rows x 2 692 B")
    ap.add_argument("--chunk-target-mb", type=float, default=0.0, help="cut a segment into files of about this size")
    ap.add_argument("--crc32", type=int, default=int(os.environ.get("LZ_SAVE_CRC32", "1")))
    ap.add_argument("--dir", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args(argv)
    made = args.dir is None
    args.dir = args.dir or tempfile.mkdtemp(prefix="lz_host_io_")
    os.makedirs(args.dir, exist_ok=True)
    ctx = mp.get_context("spawn")
    args.start_evt = ctx.Event()
    out_q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args, out_q)) for r in range(args.procs)]
    for p in procs:
        p.start()
    rows, ready = [], 0
    try:
        while ready < args.procs:                                  # imports + staging buffers are set-up, not the rehearsal
            kind, _ = out_q.get(timeout=600)
            ready += kind == "ready"
        t0 = time.perf_counter()
        args.start_evt.set()
        while len(rows) < args.procs:
            kind, payload = out_q.get(timeout=3600)
            if kind == "done":
                rows.append(payload)
        wall = time.perf_counter() - t0
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    rows.sort(key=lambda r: r["rank"])
    total_gb = sum(r["gb"] for r in rows)
    files = [f for f in os.listdir(args.dir) if f.endswith(".pt")]
    out = {"what": "host-side rehearsal of N self-play workers' shard writers (no GPU)", "procs": args.procs,
           "rate_gbps_per_proc_offered": args.rate_gbps, "gb_per_proc": round(rows[0]["gb"], 3), "crc32": bool(args.crc32),
           "segment_mb": args.segment_mb, "writers_per_proc": rows[0]["writers"],
           "aggregate_gbps": round(total_gb / wall, 3), "offered_aggregate_gbps": round(args.rate_gbps * args.procs, 3),
           "wall_s": round(wall, 3), "max_producer_stalled_s": max(r["producer_stalled_s"] for r in rows),
           "sum_producer_stalled_s": round(sum(r["producer_stalled_s"] for r in rows), 3),
           "stall_fraction_worst": round(max(r["producer_stalled_s"] / r["wall_s"] for r in rows), 4),
           "staging_bytes_total": sum(r["staging_bytes"] for r in rows), "files": len(files),
           "host_threads": args.procs * (rows[0]["writers"] + 2), "cpus": len(os.sched_getaffinity(0)), "per_proc": rows,
           "dir_fs": os.popen(f"df -P {args.dir} | tail -1").read().split()[0] if os.name == "posix" else ""}
    if made and not args.keep:
        shutil.rmtree(args.dir, ignore_errors=True)
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    main()
