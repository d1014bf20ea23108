"""CPU, world_size 2 over gloo: game sharding, per-rank seeds, trajectory gather to rank 0, checkpoint broadcast."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from liuzhou_amd.distributed import broadcast_checkpoint, gather_trajectories, split_games, worker_seed
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    return TensorSelfPlayBatch(
        state_tensors=(torch.rand((n, 11, 6, 6), generator=g) < 0.3).float(),
        legal_masks=torch.rand((n, 220), generator=g) < 0.1,
        policy_targets=torch.rand((n, 220), generator=g),
        value_targets=torch.full((n,), float(rank + 1)),
        soft_value_targets=torch.rand((n,), generator=g))


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = _batch(rank, counts[rank])
        got = gather_trajectories(mine, dst=0)
        torch.manual_seed(rank)
        model = ChessNet(**MODEL_CONFIGS["tiny"])
        broadcast_checkpoint(model, src=0)
        digest = float(sum(p.double().sum() for p in model.parameters()))
        if rank == 0:
            want = [_batch(r, counts[r]) for r in range(world)]
            ok = got is not None and got.num_samples == sum(counts)
            for f in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"):
                ok = ok and torch.equal(getattr(got, f), torch.cat([getattr(w, f) for w in want]))
            q.put(("gather", ok))
        else:
            q.put(("none", got is None))
        q.put(("digest", digest))
    finally:
        dist.destroy_process_group()


def test_split_and_seeds():
    assert split_games(10, 4) == [3, 3, 2, 2]
    assert split_games(131072, 8) == [16384] * 8
    assert sum(split_games(7, 8)) == 7 and split_games(0, 3) == [0, 0, 0]
    assert worker_seed(3, 0) == 3 * 10007 + 9973 and worker_seed(0, 7) == 8 * 9973


def test_gather_and_broadcast_world2_gloo():
    world, counts = 2, [5, 9]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    items = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = {}
    digests = []
    for k, v in items:
        if k == "digest":
            digests.append(v)
        else:
            d[k] = v
    assert d.get("gather") is True and d.get("none") is True
    assert len(digests) == 2 and abs(digests[0] - digests[1]) < 1e-9     # same weights after broadcast


def test_gather_with_empty_rank_world2_gloo():
    world, counts = 2, [4, 0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    items = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert dict((k, v) for k, v in items if k != "digest").get("gather") is True


def test_bench_spawns_its_own_ranks_and_relays_rank0_line():
    """`bench.py --gpus 2` started as a plain process (what the driver's N > 1 fallback does) must start the two ranks
    itself, run the barrier / max-over-ranks protocol and print rank 0's single JSON line.  `--dry-run` replaces the GPU
    work by a gloo group and no-op steps, so the spawning path runs on a CPU box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["LZ_BENCH_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["dry_run"] is True
    assert out["ms_per_step"] >= 2.0                      # rank 1 sleeps 2 ms per step: the MAX over ranks was taken


def test_gather_world4_with_an_empty_and_an_oversized_rank_gloo():
    """Four ranks, one without a row and one with 300x the others' (a rank whose games ran long / a straggler's backlog):
    rank 0 ends with every rank's rows in rank order, bit for bit."""
    world, counts = 4, [3, 0, 2000, 7]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    items = [q.get(timeout=180) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flags = [(k, v) for k, v in items if k != "digest"]
    assert ("gather", True) in flags and sum(1 for k, v in flags if k == "none" and v) == world - 1
    digests = [v for k, v in items if k == "digest"]
    assert len(digests) == world and max(digests) - min(digests) < 1e-9


def test_bench_dry_run_with_eight_ranks_rehearses_the_scale_run():
    """`bench.py --gpus 8 --dry-run`: the shape of the driver's 8-GPU SCALE run (spawned ranks, barrier, MAX over ranks, the
    C4 gather to rank 0 with one empty rank, per-rank rows in the line) over gloo on the CPU -- no GPU, no RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["LZ_BENCH_PORT"] = str(_free_port())
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["ms_per_step"] >= 8.0      # rank 7 sleeps 8 ms per step
    assert [p["rank"] for p in out["per_rank"]] == list(range(8)) and out["per_rank"][7]["rows"] == 0
    assert out["gather"]["rows"] == sum(p["rows"] for p in out["per_rank"]) == 7 * 16 * 4
    assert out["gather"]["source_ranks"] == list(range(7))
