"""GPU: the pipelined worker (finished-row log -> pinned staging -> chunk files while the wave goes on playing).

Reference behaviour being replaced: v1/python/self_play_worker.py:430-546 (play a chunk to the end, `.to("cpu")`, save).
The streamed shard must hold exactly the rows the runner records for the same games -- as a multiset: the log is
game-major, the runner's arena ply-major -- in files with the reference's payload / manifest keys."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _row_keys(state, legal, policy, value, soft):
    """One bytes key per row (all five tensors), sorted."""
    n = int(state.shape[0])
    cols = [state.reshape(n, -1).contiguous().numpy().view(np.uint8).reshape(n, -1),
            legal.reshape(n, -1).to(torch.uint8).numpy(),
            policy.reshape(n, -1).contiguous().numpy().view(np.uint8).reshape(n, -1),
            value.reshape(n, 1).contiguous().numpy().view(np.uint8).reshape(n, -1),
            soft.reshape(n, 1).contiguous().numpy().view(np.uint8).reshape(n, -1)]
    flat = np.concatenate(cols, axis=1)
    return sorted(bytes(r) for r in flat)


def test_log_kernel_moves_finished_rows_in_slot_order_and_applies_back_pressure():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import _lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    G, Tmax, A = 1500, 12, 220                                   # > 1 slot per scan thread and a ragged last thread
    counts = torch.randint(0, Tmax + 1, (G,), generator=g)
    done = torch.rand(G, generator=g) < 0.3
    live = (torch.rand(G * Tmax, 11, 6, 6, generator=g), torch.rand(G * Tmax, A, generator=g) < 0.5,
            torch.rand(G * Tmax, A, generator=g), torch.rand(G * Tmax, generator=g), torch.rand(G * Tmax, generator=g))
    slots = [int(s) for s in torch.nonzero(done & (counts > 0)).view(-1)]
    want_rows = [s * Tmax + j for s in slots for j in range(int(counts[s]))]

    def run(capacity, counts_dev, log, counters):
        base = torch.empty(G, dtype=torch.int64, device=dev)
        L.check(L.lib().lz_wave_log_finished(
            L.ptr(d_done), L.ptr(counts_dev), L.i64(G), L.i64(Tmax), L.i64(A), *(L.ptr(t) for t in d_live),
            *(L.ptr(t) for t in log), L.i64(capacity), L.ptr(counters), L.ptr(base), L.stream_ptr(dev)), "log")
        torch.cuda.synchronize()

    d_done = done.to(dev)
    d_live = [t.to(dev) for t in live]
    mk = lambda cap: [torch.zeros(cap, 11, 6, 6, device=dev), torch.zeros(cap, A, dtype=torch.bool, device=dev),
                      torch.zeros(cap, A, device=dev), torch.zeros(cap, device=dev), torch.zeros(cap, device=dev)]
    # 1) everything fits
    cap = len(want_rows) + 7
    log, counters, c = mk(cap), torch.zeros(4, dtype=torch.int64, device=dev), counts.to(dev)
    run(cap, c, log, counters)
    assert counters.tolist() == [len(want_rows), len(slots), 0, 0]
    idx = torch.tensor(want_rows)
    for got, src in zip(log, live):
        assert torch.equal(got[:len(want_rows)].cpu(), src[idx])
    after = counts.clone()
    after[torch.tensor(slots)] = 0
    assert torch.equal(c.cpu(), after)                            # logged slots are free, live / empty ones untouched
    # 2) a log that only holds part of them: a prefix (in slot order) goes, the rest waits and goes into the next arena
    cap = len(want_rows) // 2
    log, counters, c = mk(cap), torch.zeros(4, dtype=torch.int64, device=dev), counts.to(dev)
    run(cap, c, log, counters)
    fit, rows_fit = 0, 0
    for s in slots:
        if rows_fit + int(counts[s]) > cap:
            break
        fit, rows_fit = fit + 1, rows_fit + int(counts[s])
    assert counters.tolist() == [rows_fit, fit, len(slots) - fit, len(want_rows) - rows_fit]
    for got, src in zip(log, live):
        assert torch.equal(got[:rows_fit].cpu(), src[idx[:rows_fit]])
    assert int((c.cpu()[torch.tensor(slots[fit:])] > 0).all())   # the waiting slots keep their rows
    log2, counters2 = mk(len(want_rows)), torch.zeros(4, dtype=torch.int64, device=dev)
    run(len(want_rows), c, log2, counters2)
    assert counters2.tolist() == [len(want_rows) - rows_fit, len(slots) - fit, 0, 0]
    for got, src in zip(log2, live):
        assert torch.equal(got[:len(want_rows) - rows_fit].cpu(), src[idx[rows_fit:]])
    assert torch.equal(c.cpu(), after)


def _load_shards(tmp_path, manifest):
    parts = [torch.load(tmp_path / str(name), map_location="cpu") for name in manifest["shard_files"]]
    for p, size in zip(parts, manifest["shard_sizes"]):
        assert set(p) == {"state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets", "stats",
                          "metadata"}
        assert p["metadata"]["payload_format"] == "v1_sharded_shard" and int(p["state_tensors"].shape[0]) == int(size)
        assert p["state_tensors"].dtype == torch.float32 and p["legal_masks"].dtype == torch.bool
        # a file owns exactly its rows (torch.save writes whole storages)
        assert p["state_tensors"].untyped_storage().nbytes() == p["state_tensors"].numel() * 4
    cat = lambda k: torch.cat([p[k] for p in parts])
    return tuple(cat(k) for k in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"))


@pytest.mark.parametrize("log_rows", [0, 64])                   # 64 rows per arena: two games fill it -> back-pressure
def test_streamed_tree_shard_holds_exactly_the_runners_rows(tmp_path, monkeypatch, log_rows):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.self_play_worker import run_self_play_worker
    from liuzhou_amd.tree_engine import clear_engine_cache, self_play_tree_gpu
    monkeypatch.setenv("LZ_WORKER_SEGMENT_GAMES", "5")
    if log_rows:
        monkeypatch.setenv("LZ_WORKER_LOG_ROWS", str(log_rows))
    torch.manual_seed(11)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    state_path = tmp_path / "model_state.pt"
    torch.save(model.state_dict(), state_path)
    games, slots, sims, plies, seed = 27, 8, 6, 30, 5
    common = dict(mcts_simulations=sims, temperature_init=1.0, temperature_final=0.1, temperature_threshold=4,
                  exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, soft_value_k=2.0,
                  opening_random_moves=2, max_game_plies=plies)
    manifest_path = tmp_path / "worker.pt"
    row = run_self_play_worker(worker_idx=0, shard_device="cuda:0", shard_games=games, seed=seed,
                               model_state_path=str(state_path), output_path=str(manifest_path),
                               concurrent_games_per_device=slots, chunk_output_dir=str(tmp_path),
                               chunk_file_prefix="w", search_backend="portable", **common)
    manifest = torch.load(manifest_path, map_location="cpu")
    got = _load_shards(tmp_path, manifest)
    counters = manifest["stats"]["mcts_counters"]
    assert counters["stream_segments"] >= 4 and manifest["num_shards"] == len(manifest["shard_files"]) >= 4
    assert manifest["metadata"]["num_selfplay_batches"] == counters["stream_segments"]
    if log_rows:
        assert counters["stream_blocked_polls"] >= 0
    clear_engine_cache()
    # the same games straight from the runner (the worker's RNG key for its first run; per-game counter RNG: a game's
    # noise and picks depend on (seed, game id, ply) only, not on the slot or the moment it is played)
    twin = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    twin.load_state_dict(model.state_dict())
    net = FusedNet(twin.to("cuda:0"), torch.device("cuda:0"))
    batch, st = self_play_tree_gpu(net, num_games=games, concurrent_games=slots, device="cuda:0", add_dirichlet_noise=True,
                                   sample_moves=True, seed=(seed * 1000003 + 1) & 0x7FFFFFFFFFFFFFFF, **common)
    clear_engine_cache()
    want = tuple(t.cpu() for t in (batch.state_tensors, batch.legal_masks, batch.policy_targets, batch.value_targets,
                                   batch.soft_value_targets))
    assert row["num_samples"] == manifest["num_samples"] == batch.num_samples == int(st.num_positions)
    assert manifest["stats"]["num_games"] == games and manifest["stats"]["num_positions"] == batch.num_samples
    assert _row_keys(*got) == _row_keys(*want)
    assert torch.isfinite(got[3]).all() and torch.isfinite(got[4]).all()
    for k in ("black_wins", "white_wins", "draws"):
        assert manifest["stats"][k] == getattr(st, k)


def test_streamed_root_shard_invariants_and_classic_loop_still_works(tmp_path, monkeypatch):
    """Variant R (the reference's default backend): streamed shard vs the chunk loop (LZ_WORKER_STREAM=0) -- same number of
    games, finite targets, policies on the legal set; the reference-style loop emits one file per chunk."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_worker import run_self_play_worker
    torch.manual_seed(12)
    state_path = tmp_path / "model_state.pt"
    torch.save(ChessNet(**MODEL_CONFIGS["b6c64"]).state_dict(), state_path)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LZ_WORKER_STREAM", mode)
        monkeypatch.setenv("LZ_WORKER_SEGMENT_GAMES", "16")
        d = tmp_path / f"m{mode}"
        os.makedirs(d)
        run_self_play_worker(worker_idx=1, shard_device="cuda:0", shard_games=96, seed=3, model_state_path=str(state_path),
                             output_path=str(d / "worker.pt"), mcts_simulations=16, temperature_init=1.0,
                             temperature_final=0.1, temperature_threshold=6, exploration_weight=1.0, dirichlet_alpha=0.3,
                             dirichlet_epsilon=0.25, soft_value_k=2.0, opening_random_moves=0, max_game_plies=40,
                             concurrent_games_per_device=32, chunk_output_dir=str(d), chunk_file_prefix="w",
                             search_backend="cuda_root")
        m = torch.load(d / "worker.pt", map_location="cpu")
        state, legal, policy, value, soft = _load_shards(d, m)
        assert m["stats"]["num_games"] == 96 and m["num_samples"] == state.shape[0] == m["stats"]["num_positions"]
        assert torch.isfinite(value).all() and torch.isfinite(soft).all()
        assert torch.allclose(policy.sum(1), torch.ones(policy.shape[0]), atol=1e-4)
        assert bool((policy[~legal] == 0).all())
        assert m["avg_bytes_per_sample"] == 2692
        out[mode] = m
    assert out["0"]["num_shards"] == 3 and "stream_segments" not in out["0"]["stats"]["mcts_counters"]
    assert out["1"]["num_shards"] >= 3 and out["1"]["stats"]["mcts_counters"]["stream_segments"] >= out["1"]["num_shards"]
    assert set(out["0"]["metadata"]) == set(out["1"]["metadata"]) and set(out["0"]) == set(out["1"])


def test_worker_falls_back_to_the_chunk_loop_when_the_slot_major_arena_does_not_fit(tmp_path, monkeypatch):
    """The streamed worker's live arena is slots x max_game_plies rows whatever the game lengths (22.6 GB at 16 384 x 512);
    when that is more than its share of the free device memory the worker plays the reference's chunk loop instead of
    failing in its first ply, and says so in the manifest."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_worker import run_self_play_worker, streaming_footprint
    fp = streaming_footprint(16384, 512)
    assert 22.5e9 < fp["live_arena_bytes"] < 22.7e9 and fp["device_bytes"] == fp["live_arena_bytes"] + fp["log_arena_bytes"]
    assert fp["pinned_host_bytes"] == 2 * fp["log_arena_bytes"]
    torch.manual_seed(12)
    state_path = tmp_path / "model_state.pt"
    torch.save(ChessNet(**MODEL_CONFIGS["b6c64"]).state_dict(), state_path)
    monkeypatch.setenv("LZ_WORKER_STREAM", "1")
    monkeypatch.setenv("LZ_WORKER_STREAM_SHARE", "1e-9")
    run_self_play_worker(worker_idx=0, shard_device="cuda:0", shard_games=6, seed=3, model_state_path=str(state_path),
                         output_path=str(tmp_path / "worker.pt"), mcts_simulations=4, temperature_init=1.0,
                         temperature_final=0.1, temperature_threshold=6, exploration_weight=1.0, dirichlet_alpha=0.3,
                         dirichlet_epsilon=0.25, soft_value_k=2.0, opening_random_moves=0, max_game_plies=24,
                         concurrent_games_per_device=3, chunk_output_dir=str(tmp_path), chunk_file_prefix="w",
                         search_backend="cuda_root")
    m = torch.load(tmp_path / "worker.pt", map_location="cpu")
    assert m["metadata"]["streamed"] is False and "chunk loop" in m["metadata"]["stream_fallback"]
    assert m["num_shards"] == 2 and "stream_segments" not in m["stats"]["mcts_counters"] and m["stats"]["num_games"] == 6
