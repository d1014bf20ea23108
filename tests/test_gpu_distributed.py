"""GPU, world_size 2: the compact 360-byte trajectory gather (`distributed._gather_compact`: the path RCCL takes
between the GPUs of a node) and the flat checkpoint broadcast under a real process group.  Both ranks share cuda:0
(a 1-GPU box); RCCL refuses two ranks on one device, so the group is gloo and the records are staged through host
memory -- the count / bad-row protocol and the pack / unpack kernels are exactly those of the RCCL path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
FIELDS = ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _play(rank, games, plies):
    """Real trajectory rows: a few plies of root-PUCT self-play on the HIP operators (tiny net), on cuda:0."""
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to("cuda:0")
    torch.manual_seed(100 + rank)
    batch, _ = self_play_v1_gpu(model, num_games=games, mcts_simulations=8, temperature_init=1.0, temperature_final=0.1,
                                temperature_threshold=10, exploration_weight=1.0, device="cuda:0",
                                add_dirichlet_noise=True, sample_moves=True, concurrent_games=games,
                                max_game_plies=plies, autocast_dtype="float32")
    return batch


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from liuzhou_amd.distributed import broadcast_checkpoint, gather_trajectories
        from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
        from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
        mine = _play(rank, games=3 + 2 * rank, plies=10 + rank) if mode != "empty" or rank == 0 else _play(rank, 1, 1)
        if mode == "empty" and rank == 1:
            mine = TensorSelfPlayBatch(*(getattr(mine, f)[:0] for f in FIELDS))
        if mode == "bad" and rank == 1:                         # policy mass on an illegal action: not representable
            pol = mine.policy_targets.clone()
            illegal = (~mine.legal_masks[0]).nonzero().view(-1)[0]
            pol[0, illegal] = 0.25
            mine = TensorSelfPlayBatch(mine.state_tensors, mine.legal_masks, pol, mine.value_targets, mine.soft_value_targets)
        raised = None
        try:
            got = gather_trajectories(mine, dst=0, compact=True)
        except RuntimeError as exc:
            raised, got = str(exc), None
        if mode == "bad":
            q.put((rank, "raised", raised is not None and "not representable" in raised))
            return
        assert raised is None, raised
        # the reference layout of the same rows, through the five-tensor path on host tensors
        cpu = TensorSelfPlayBatch(*(getattr(mine, f).cpu() for f in FIELDS))
        want = gather_trajectories(cpu, dst=0, compact=False)
        if rank == 0:
            ok = got is not None and got.num_samples == want.num_samples and got.state_tensors.is_cuda
            for f in FIELDS:
                a, b = getattr(got, f).cpu(), getattr(want, f)
                same = torch.equal(a, b) if a.dtype == torch.bool else torch.equal(a.view(torch.int32), b.view(torch.int32))
                ok = ok and same
            q.put((rank, "gather", bool(ok)))
        else:
            q.put((rank, "gather", got is None))
        torch.manual_seed(rank)
        model = ChessNet(**MODEL_CONFIGS["b6c64"]).to("cuda:0")
        broadcast_checkpoint(model.cpu(), src=0)                 # gloo: host tensors (RCCL broadcasts device tensors)
        q.put((rank, "digest", float(sum(p.double().sum() for p in model.parameters())) +
               float(sum(b.double().sum() for b in model.buffers()))))
    finally:
        dist.destroy_process_group()


def _run(mode, n_items):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    items = [q.get(timeout=300) for _ in range(n_items)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return items


def test_compact_gather_under_a_process_group_is_bit_exact():
    items = _run("normal", 4)
    assert all(v is True for _, k, v in items if k == "gather") and sum(k == "gather" for _, k, _ in items) == 2
    d = [v for _, k, v in items if k == "digest"]
    assert len(d) == 2 and abs(d[0] - d[1]) < 1e-9


def test_compact_gather_with_an_empty_rank():
    items = _run("empty", 4)
    assert all(v is True for _, k, v in items if k == "gather")


def test_unrepresentable_row_makes_every_rank_raise():
    items = _run("bad", 2)
    assert sorted(r for r, _, _ in items) == [0, 1] and all(v is True for _, _, v in items)


def test_bench_two_ranks_with_gather_in_the_timed_region():
    """`bench.py --gpus 2` -- with more than one rank the gather is part of the measured config by default (C4's code path: games sharded over the ranks, barrier + max-over-ranks timing,
    the compact gather of the timed steps' rows inside the timed region) started as a plain process: it spawns its two
    ranks itself.  Both ranks share cuda:0 here (gloo group; RCCL refuses two ranks per device), small workload."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(LZ_BENCH_BACKEND="gloo", LZ_BENCH_SHARE_GPU="1", LZ_BENCH_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--games", "256", "--sims", "16", "--model", "b6c64", "--soak-seconds", "0",
                        "--reuse-factor", "2", "--no-probe"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 256 * 4 / (out["ms_per_step"] * 4e-3)) < 2e-3 * out["value"]      # ms_per_step is rounded
    g = out["gather"]
    assert g["rows_on_rank0"] == 2 * 256 * 4 and g["record_bytes"] == 360 and g["bytes_received"] == 256 * 4 * 360
    assert "gathered to rank 0" in out["config"]["workload"] and out["config"]["workload"].startswith("steady-state harness, C4")
    # per-rank rows (all-gathered): a SCALE run can attribute a loss to a straggler, the gather or a power-capped package
    pr = out["per_rank"]
    assert [d["rank"] for d in pr] == [0, 1] and out["backend"] == "gloo" and out["rccl_ranks"] == 2
    for d in pr:
        assert 0 < d["play_ms_per_step"] <= d["ms_per_step"] * 1.001 and d["leaf_evals"] > 0 and d["gather_ms"] > 0
        assert d["pci"].count(":") == 2
    assert out["distinct_devices"] == 1                             # both ranks share cuda:0 here; under RCCL bench.py raises
    assert abs(max(d["ms_per_step"] for d in pr) - out["ms_per_step"]) < 0.01 * out["ms_per_step"] + 0.01
    assert out["per_rank_summary"]["straggler_ratio"] >= 1.0


def _staged_loop(world, overlap, iterations=3, games=48, plies=6, extra=()):
    """scripts/staged_loop.py with `world` ranks on cuda:0 (gloo group, records / checkpoint staged through host memory):
    ranks 0..world-2 play, rank world-1 trains and does not play."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(LZ_DIST_BACKEND="gloo", LZ_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "scripts", "staged_loop.py"), "--iterations", str(iterations), "--games-per-gpu", str(games),
           "--sims", "8", "--max-game-plies", str(plies), "--batch-size", "128", "--overlap", str(overlap), *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # one JSON line, from the trainer rank
    return json.loads(lines[0])


@pytest.mark.parametrize("world,overlap", [(2, 0), (3, 0), (3, 1)])
def test_staged_loop_multi_rank_trainer_does_not_play(world, overlap):
    """C5's world > 1 branches: the trainer rank joins the gather with an empty batch and is its destination, every player
    re-packs the broadcast checkpoint into its live FusedNet buffers.  A game cannot end within 6 plies (placement
    phase), so every player contributes exactly games x plies rows per iteration."""
    games, plies, its = 48, 6, 3
    out = _staged_loop(world, overlap, iterations=its, games=games, plies=plies)
    assert out["world_size"] == world and out["trainer_rank"] == world - 1 and out["backend"] == "gloo"
    assert out["player_ranks"] == list(range(world - 1)) and out["overlap"] == overlap
    log = out["iterations"]
    assert len(log) == its
    for e in log:
        assert e["positions"] == (world - 1) * games * plies
        assert e["weights_equal_on_all_ranks"] is True           # players' packed device weights == the trainer's model
    digests = [e["weights_digest"] for e in log]
    if overlap:
        # lag-1: iteration 1 has nothing to train on yet (hand-off of the initial weights), then generation i-1 each time
        assert log[0]["train_samples"] == 0 and all(e["train_samples"] == (world - 1) * games * plies for e in log[1:])
        assert len(set(digests[1:])) == its - 1 and digests[1] != digests[0]
        assert out["tail"]["train_samples"] == (world - 1) * games * plies and out["tail"]["avg_loss"] is not None
    else:
        assert all(e["train_samples"] == (world - 1) * games * plies for e in log)
        assert len(set(digests)) == its                           # the checkpoint changes every iteration
        assert out["tail"] is None
    assert all(e["avg_loss"] is not None for e in log if e["train_samples"])
    assert out["steady_state_positions_per_sec"] > 0


def test_rccl_group_of_one_runs_the_device_tensor_collectives():
    """Every multi-rank test above is gloo (two ranks cannot share a device under RCCL).  This one initialises a
    world-size-1 `nccl` group -- RCCL library load, communicator creation -- and runs the device-tensor branch of
    `gather_trajectories` (counts all-gather) and `broadcast_checkpoint` plus bench.py's all-reduce through it
    (tests/nccl_world1.py, in a child process with a time limit: a hung collective must not hang the suite)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        res = subprocess.run([sys.executable, os.path.join(root, "tests", "nccl_world1.py"), str(_free_port())],
                             capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired as exc:
        pytest.fail(f"the world-size-1 RCCL job did not finish in 240 s: {exc.stdout} {exc.stderr}")
    assert res.returncode == 0, res.stdout + res.stderr
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["gather_bit_exact"] is True and out["rows"] > 0
    assert out["broadcast_keeps_weights"] is True and out["all_reduce_max"] == 3.25 and out["barrier"] is True
