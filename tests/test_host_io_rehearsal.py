"""CPU: the host side of N self-play workers on one node, rehearsed without GPUs (scripts/rehearse_host_io.py): N processes
drive the real `StreamedShardFiles` writer into ONE directory at an offered rate; and the rule that sizes a worker's
writer pool from its share of the node's cores (`self_play_worker.default_writer_threads`)."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_four_concurrent_shard_writers_one_directory(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rehearse_host_io.py"), "--procs", "4",
                          "--gb-per-proc", "0.02", "--rate-gbps", "0.5", "--segment-mb", "4", "--chunk-target-mb", "1.5",
                          "--dir", str(tmp_path), "--keep"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["procs"] == 4 and len(d["per_proc"]) == 4 and d["aggregate_gbps"] > 0
    assert 0.0 <= d["stall_fraction_worst"] <= 1.0 and d["host_threads"] == 4 * (d["writers_per_proc"] + 2)
    rows = 0
    for r in d["per_proc"]:
        man = torch.load(tmp_path / f"worker_{r['rank']:02d}.pt", weights_only=False)
        assert man["payload_format"] == "v1_worker_chunk_manifest" and man["num_samples"] == r["rows"] == sum(man["shard_sizes"])
        assert len(man["shard_files"]) == r["files"] >= 5                     # 4 MB segments cut into ~1.5 MB files
        for name, size in zip(man["shard_files"], man["shard_sizes"]):
            assert name.startswith(f"it000.w{r['rank']:02d}.chunk")           # no two workers share a file name
            shard = torch.load(tmp_path / name, weights_only=False)
            assert shard["state_tensors"].shape == (size, 11, 6, 6) and shard["legal_masks"].dtype == torch.bool
            assert shard["state_tensors"].untyped_storage().nbytes() == size * 11 * 36 * 4   # a file owns exactly its rows
        rows += man["num_samples"]
    assert rows == sum(r["rows"] for r in d["per_proc"]) and d["files"] == sum(r["files"] for r in d["per_proc"]) + 4


def test_writer_pool_follows_the_workers_share_of_the_cores(monkeypatch):
    from liuzhou_amd import self_play_worker as W
    for k in ("LZ_WORKER_WRITERS", "LZ_WORKERS_ON_NODE", "LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(128)))
    assert W.workers_on_node() == 1 and W.default_writer_threads() == 3
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert W.workers_on_node() == 8 and W.default_writer_threads() == 3       # 16 cores per worker
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(32)))
    assert W.default_writer_threads() == 2                                     # 4 cores per worker: player + copier + 2
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(16)))
    assert W.default_writer_threads() == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")                                # torchrun: ranks on THIS node
    assert W.workers_on_node() == 2 and W.default_writer_threads() == 3
    monkeypatch.setenv("LZ_WORKERS_ON_NODE", "16")                             # the stage's own count wins
    assert W.default_writer_threads() == 1
    monkeypatch.setenv("LZ_WORKER_WRITERS", "5")                               # an explicit setting wins over the rule
    assert W.default_writer_threads() == 5


def test_files_saved_without_crc32_load_like_the_others(tmp_path):
    """LZ_SAVE_CRC32=0 halves a writer thread's work (scripts/rehearse_host_io.py); `torch.load` -- the reference's loader --
    reads such files unchanged."""
    from liuzhou_amd.self_play_worker import configure_save
    t = {"state_tensors": torch.arange(4096, dtype=torch.float32).reshape(64, 64), "metadata": {"a": 1}}
    try:
        assert configure_save(crc32=False) is False
        torch.save(t, tmp_path / "nocrc.pt")
    finally:
        assert configure_save(crc32=True) is True
    torch.save(t, tmp_path / "crc.pt")
    a, b = torch.load(tmp_path / "nocrc.pt", weights_only=False), torch.load(tmp_path / "crc.pt", weights_only=False)
    assert torch.equal(a["state_tensors"], b["state_tensors"]) and a["metadata"] == b["metadata"]
