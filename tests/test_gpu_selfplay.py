"""GPU: the root-PUCT self-play wave loop on the HIP operators reproduces the reference's own trace
(tests/golden/g8_selfplay.npz: v1 runner, 4 games x 32 sims, tiny net, deterministic picks)."""
import numpy as np
import pytest
import torch

from tests.golden_utils import load

pytestmark = pytest.mark.gpu


def test_gpu_root_selfplay_matches_reference_trace():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    z = load("g8_selfplay.npz")
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to("cuda:0")
    batch, stats = self_play_v1_gpu(model, num_games=4, mcts_simulations=32, temperature_init=1.0,
                                    temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                    device="cuda:0", add_dirichlet_noise=False, soft_value_k=2.0,
                                    opening_random_moves=0, max_game_plies=512, sample_moves=False,
                                    concurrent_games=4, autocast_dtype="float32", collect_step_timing=True)
    n = int(z["num_positions"])
    assert stats.num_positions == n
    want_states = np.unpackbits(z["state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(batch.state_tensors.cpu().numpy(), want_states)
    assert np.array_equal(batch.legal_masks.cpu().numpy(), np.unpackbits(z["legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(batch.policy_targets.cpu().numpy(), z["policy_targets"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(batch.value_targets.cpu().numpy(), z["value_targets"])
    np.testing.assert_allclose(batch.soft_value_targets.cpu().numpy(), z["soft_value_targets"], atol=1e-6, rtol=0)
    assert (stats.black_wins, stats.white_wins, stats.draws) == (int(z["black_wins"]), int(z["white_wins"]), int(z["draws"]))
    assert abs(stats.avg_game_length - float(z["avg_game_length"])) < 1e-6
    assert set(stats.step_timing_ms) == {"root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms"}


@pytest.mark.parametrize("tag", ["p2k4", "p3k3"])
def test_gpu_root_selfplay_with_topk_lookahead_matches_reference_trace(tag):
    """sparse_ply = 2 / 3 (the reference's experimental multi-ply refinement of the children's values,
    v1/python/mcts_gpu.py:976-1160) on the HIP operators reproduces the reference runner's own traces (g12)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    z = load("g12_sparse_selfplay.npz")
    ply, top_k, games, sims, max_plies = (int(x) for x in z[f"{tag}_config"])
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to("cuda:0")
    batch, stats = self_play_v1_gpu(model, num_games=games, mcts_simulations=sims, temperature_init=1.0,
                                    temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                    device="cuda:0", add_dirichlet_noise=False, soft_value_k=2.0,
                                    opening_random_moves=0, max_game_plies=max_plies, sample_moves=False,
                                    concurrent_games=games, autocast_dtype="float32", sparse_ply=ply,
                                    sparse_top_k=top_k)
    n = z[f"{tag}_policy_targets"].shape[0]
    assert stats.num_positions == n
    want_states = np.unpackbits(z[f"{tag}_state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(batch.state_tensors.cpu().numpy(), want_states)
    assert np.array_equal(batch.legal_masks.cpu().numpy(),
                          np.unpackbits(z[f"{tag}_legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(batch.policy_targets.cpu().numpy(), z[f"{tag}_policy_targets"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(batch.value_targets.cpu().numpy(), z[f"{tag}_value_targets"])
    np.testing.assert_allclose(batch.soft_value_targets.cpu().numpy(), z[f"{tag}_soft_value_targets"], atol=1e-6, rtol=0)
    assert [stats.black_wins, stats.white_wins, stats.draws] == z[f"{tag}_outcome"].tolist()


def test_gpu_selfplay_sampled_contract():
    """Contract asserts of tests/v1/test_v1_tensor_pipeline_smoke.py:73-111 (noise + sampling + opening moves)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    torch.manual_seed(0)
    model = ChessNet().eval().to("cuda:0")
    samples, stats = self_play_v1_gpu(model=model, num_games=3, mcts_simulations=4, temperature_init=1.0,
                                      temperature_final=0.1, temperature_threshold=4, exploration_weight=1.0,
                                      device="cuda:0", add_dirichlet_noise=True, dirichlet_alpha=0.3,
                                      dirichlet_epsilon=0.25, soft_value_k=2.0, opening_random_moves=2,
                                      max_game_plies=96, sample_moves=True, concurrent_games=2)
    assert stats.num_games == 3 and samples.num_samples > 0
    assert samples.state_tensors.shape[1:] == (11, 6, 6)
    assert samples.legal_masks.shape == (samples.num_samples, 220)
    assert samples.policy_targets.shape == (samples.num_samples, 220)
    assert samples.value_targets.shape == (samples.num_samples,)
    assert int(stats.mcts_counters.get("forced_uniform_pick_count", 0)) > 0
    pol = samples.policy_targets
    assert torch.allclose(pol.sum(1), torch.ones_like(pol.sum(1)), atol=1e-4)
    assert bool((pol[~samples.legal_masks] == 0).all())
    assert bool(torch.isfinite(samples.value_targets).all()) and bool(torch.isfinite(samples.soft_value_targets).all())
    assert bool((samples.value_targets.abs() <= 1).all())


def test_tree_self_play_runner_dual_stream_matches_contract():
    """self_play_tree_gpu with the two-stream search: same tensor contract, every game played to the ply cap."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import self_play_tree_gpu
    torch.manual_seed(2)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0"))
    for dual in (True, False):
        batch, stats = self_play_tree_gpu(net, num_games=65, mcts_simulations=12, temperature_init=1.0,
                                          temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                          device="cuda:0", max_game_plies=30, concurrent_games=65, dual_stream=dual)
        assert batch.num_samples == 65 * 30 == stats.num_positions
        assert torch.allclose(batch.policy_targets.sum(1), torch.ones(batch.num_samples, device="cuda:0"), atol=1e-4)
        assert bool((batch.policy_targets[~batch.legal_masks] == 0).all())
        assert bool(torch.isfinite(batch.value_targets).all())
        assert stats.mcts_counters["leaf_eval_count"] == 65 * 13 * 30
    # 100 games through 40 slots: finished slots start the remaining games at once (60, then a partial batch of 20);
    # the host-side wave loop (no device tail) plays the same number of positions in three sequential waves
    for tail in (True, False):
        batch, stats = self_play_tree_gpu(net, num_games=100, mcts_simulations=8, temperature_init=1.0,
                                          temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                          device="cuda:0", max_game_plies=24, concurrent_games=40, device_tail=tail)
        assert batch.num_samples == 100 * 24 == stats.num_positions and stats.avg_game_length == 24.0
        assert stats.black_wins + stats.white_wins + stats.draws == 100
        assert bool(torch.isfinite(batch.value_targets).all()) and bool(torch.isfinite(batch.soft_value_targets).all())
        assert stats.mcts_counters["leaf_eval_count"] == 120 * 9 * 24      # 72 plies x all 40 slots x (8 sims + root)


def test_tree_runner_engine_cache_gives_identical_games_and_follows_refreshed_weights():
    """self_play_tree_gpu keeps its engine + captured graphs between calls with the same network buffers and shape (the
    worker calls it per chunk, the staged loop per iteration): a second call is a cache hit and plays bit-identical games
    for the same seed, other games for another seed; `FusedNet.refresh` writes a new checkpoint into the same buffers, so
    the cached graph searches with the NEW weights; another shape or LZ_ENGINE_CACHE=0 builds a fresh engine."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import tree_engine as TE
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    TE.clear_engine_cache()
    torch.manual_seed(3)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0")
    net = FusedNet(model)
    kw = dict(num_games=48, mcts_simulations=10, temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
              exploration_weight=1.0, device="cuda:0", max_game_plies=12, concurrent_games=32)
    fields = ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")
    same = lambda a, b: all(torch.equal(getattr(a, f), getattr(b, f)) for f in fields)
    b1, s1 = TE.self_play_tree_gpu(net, seed=77, **kw)
    b2, s2 = TE.self_play_tree_gpu(net, seed=77, **kw)
    b3, s3 = TE.self_play_tree_gpu(net, seed=78, **kw)
    assert (s1.mcts_counters["engine_cache_hit"], s2.mcts_counters["engine_cache_hit"], s3.mcts_counters["engine_cache_hit"]) == (0, 1, 1)
    assert b1.num_samples == b2.num_samples == 48 * 12 and same(b1, b2) and not same(b1, b3)
    assert s2.mcts_counters["leaf_eval_count"] == s1.mcts_counters["leaf_eval_count"]
    for st in (s1, s2, s3):                                      # nothing the bounded arenas did to these runs
        assert (st.mcts_counters["edge_pool_refused"], st.mcts_counters["reuse_dropped"], st.mcts_counters["reuse_pruned"]) == (0, 0, 0)
    engine = next(iter(TE._ENGINE_CACHE.values()))
    assert engine.engine.pool_status()["refused_expansions"] == 0
    # new weights into the same device buffers: the cached engine must play what a fresh engine plays with them
    torch.manual_seed(4)
    model2 = ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0")
    net.refresh(model2)
    b4, s4 = TE.self_play_tree_gpu(net, seed=77, **kw)
    assert s4.mcts_counters["engine_cache_hit"] == 1 and not same(b1, b4)
    TE.clear_engine_cache()
    b5, s5 = TE.self_play_tree_gpu(FusedNet(model2), seed=77, **kw)
    assert s5.mcts_counters["engine_cache_hit"] == 0 and same(b4, b5)
    # another shape evicts (limit 1), the old shape then misses
    TE.self_play_tree_gpu(net, seed=77, **{**kw, "concurrent_games": 16})
    assert len(TE._ENGINE_CACHE) == 1
    TE.clear_engine_cache()
    assert len(TE._ENGINE_CACHE) == 0


@pytest.mark.parametrize("sims,use_graph", [(1, False), (64, False), (200, True)])
def test_fused_root_search_equals_operator_chain(sims, use_graph):
    """FusedRootSearch (two fixed-shape kernels around the network launches, no host sync) == V1RootMCTS.search_batch
    (the reference's operator chain) on mid-game and terminal positions: same policy, picks, masks, values."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.mcts_gpu import V1RootMCTS, V1RootMCTSConfig
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.root_search_fused import FusedRootSearch
    from tests.golden_utils import states, FIELDS
    from tests.tree_parity import to_gpu_batch
    dev = "cuda:0"
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    z = load("g1_rules.npz"); z2 = load("g2_edges.npz")
    st = states(z, "s")
    B = 700
    idx = np.random.default_rng(sims).integers(0, st["board"].shape[0], B)
    sub = {f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}
    batch = to_gpu_batch(sub, dev)
    # a few roots without any legal action (mark selection with nothing left to mark): terminal rows of the search
    batch.phase[:5] = 2; batch.pending_marks_remaining[:5] = 0
    cfg = V1RootMCTSConfig(num_simulations=sims, exploration_weight=1.25, add_dirichlet_noise=False, sample_moves=False)
    ref = V1RootMCTS(model=net, config=cfg, device=torch.device(dev))
    temps = torch.where(torch.arange(B, device=dev) % 2 == 0, 1.0, 0.1)
    a = ref.search_batch(batch, temperatures=temps)
    fused = FusedRootSearch(net, B, sims, dev, exploration_weight=1.25, add_dirichlet_noise=False, sample_moves=False,
                            use_graph=use_graph)
    for rep in range(2):                                   # second call replays the graph
        b = fused.search_batch(batch, temperatures=temps)
        assert torch.equal(a.terminal_mask, b.terminal_mask) and bool(a.terminal_mask[:5].all())
        assert torch.equal(a.chosen_valid_mask, b.chosen_valid_mask)
        assert torch.equal(a.chosen_action_indices, b.chosen_action_indices)
        assert torch.equal(a.chosen_action_codes, b.chosen_action_codes)
        assert torch.equal(a.policy_dense, b.policy_dense)
        assert torch.allclose(a.root_value, b.root_value, atol=1e-6)
        assert torch.equal(a.legal_mask, b.legal_mask) and torch.equal(a.model_input, b.model_input)
    assert fused.children_evaluated() == int(a.legal_mask.sum().item())


@pytest.mark.parametrize("ply,top_k,mode", [(2, 4, "value_only"), (3, 3, "value_only"), (2, 8, "full"), (1, 8, "full")])
def test_fused_root_search_with_topk_lookahead_equals_operator_chain(ply, top_k, mode):
    """`sparse_ply` 2 / 3 and `child_eval_mode="full"` inside the fused root search (round 6: lz_root_topk_children /
    lz_root_refine_topk around one more prepare / collect round, all in the captured launch sequence) == the operator
    chain of V1RootMCTS, which reproduces the reference runner's own sparse traces (g12, test above): same visits-derived
    policy, picks, values -- with injected root noise, on mid-game positions, positions whose best children end the
    game or have no reply, and roots with fewer legal actions than top_k."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.mcts_gpu import V1RootMCTS, V1RootMCTSConfig
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.root_search_fused import FusedRootSearch, CAP
    from tests.golden_utils import states, FIELDS
    from tests.tree_parity import to_gpu_batch
    dev = "cuda:0"
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    st = states(load("g1_rules.npz"), "s")
    st2 = states(load("g2_edges.npz"), "s")
    B = 400
    idx = np.random.default_rng(ply * 10 + top_k).integers(0, st["board"].shape[0], B - 40)
    sub = {f: np.concatenate([np.asarray(st[f])[idx], np.resize(np.asarray(st2[f]), (40,) + np.asarray(st2[f]).shape[1:])])
           for f in FIELDS}
    batch = to_gpu_batch({f: np.ascontiguousarray(v) for f, v in sub.items()}, dev)
    sims = 64
    cfg = V1RootMCTSConfig(num_simulations=sims, exploration_weight=1.1, add_dirichlet_noise=True, sample_moves=False,
                           sparse_ply=ply, sparse_top_k=top_k, child_eval_mode=mode)
    ref = V1RootMCTS(model=net, config=cfg, device=torch.device(dev))
    temps = torch.where(torch.arange(B, device=dev) % 2 == 0, 1.0, 0.1)
    # the same Gamma draws for both: left-packed rows of the operator chain = the first `count` slots of the fused rows
    g = torch.Generator(device="cpu").manual_seed(3)
    noise72 = torch._standard_gamma(torch.full((B, CAP), 0.3), generator=g).to(dev)
    first = ref.search_batch(batch, temperatures=temps, add_dirichlet_noise=False)        # row layout of this batch
    roots = torch.nonzero(~first.terminal_mask).view(-1)
    M = int(first.legal_mask.sum(1).max().item())
    ref.injected_noise = noise72.index_select(0, roots)[:, :M]
    a = ref.search_batch(batch, temperatures=temps)
    fused = FusedRootSearch(net, B, sims, dev, exploration_weight=1.1, add_dirichlet_noise=True, sample_moves=False,
                            sparse_ply=ply, sparse_top_k=top_k, child_eval_mode=mode)
    for rep in range(2):                                   # second call replays the graph
        b = fused.search_batch(batch, temperatures=temps, injected_noise=noise72)
        assert torch.equal(a.terminal_mask, b.terminal_mask)
        assert torch.equal(a.chosen_valid_mask, b.chosen_valid_mask)
        assert torch.equal(a.chosen_action_indices, b.chosen_action_indices)
        assert torch.equal(a.policy_dense, b.policy_dense)
        assert torch.allclose(a.root_value, b.root_value, atol=1e-6)
    if ply > 1:
        picked = int((fused.top_slot >= 0).sum().item())
        assert 0 < picked < B * top_k                               # some roots have fewer legal actions than top_k
        if ply == 2:
            assert int(fused.l3_total.item()) == int(fused.l2_counts.sum().item()) > 0
    # network evaluations of the two searches: roots, children, and per lookahead round the L2 positions + their children
    l3 = int(fused.l3_total.item()) if ply > 1 else 0
    assert fused.leaf_evals == 2 * (B + fused.children_evaluated() + (ply - 1) * B * top_k + l3)


def test_steady_state_root_population_fused_equals_operator_chain():
    """The steady-state root-PUCT driver gives the same trajectory rows with the fused search as with the operator chain."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.mcts_gpu import V1RootMCTSConfig
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.steady_state import SteadyStateRootSelfPlay
    dev = "cuda:0"
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    cfg = V1RootMCTSConfig(num_simulations=32, add_dirichlet_noise=False, sample_moves=False)
    # [0] fused search + device tail (record / move / finalise / re-seat without the host), [1] operator chain + host
    # bookkeeping, [2] operator chain + device tail
    # ([2] starts with room for two plies only: the arena grows under the device cursor several times)
    # ([3]: the fused search as two halves on two streams, DualStreamRootSearch)
    pops = [SteadyStateRootSelfPlay(net, 300, cfg, dev, seed=5, max_game_plies=40, fused_search=f, device_tail=t,
                                    arena_rows=r, dual_stream=d)
            for f, t, r, d in ((True, True, None, False), (False, False, None, False), (False, True, 600, False),
                               (True, True, None, True))]
    assert pops[3].dual_stream and not pops[0].dual_stream
    assert pops[0].fused is not None and pops[1].fused is None
    assert pops[0].tail is not None and pops[1].tail is None and pops[2].tail is not None
    for p in pops:
        p.preroll(30)
        for _ in range(25):
            p.step()
    a, b = pops[0].buffer.build(), pops[1].buffer.build()
    assert a.num_samples == b.num_samples == 300 * 25
    assert torch.equal(a.state_tensors, b.state_tensors) and torch.equal(a.legal_masks, b.legal_masks)
    assert torch.equal(a.policy_targets, b.policy_targets)
    assert torch.equal(torch.nan_to_num(a.value_targets, nan=9.0), torch.nan_to_num(b.value_targets, nan=9.0))
    c = pops[2].buffer.build()
    assert torch.equal(c.state_tensors, b.state_tensors) and torch.equal(c.policy_targets, b.policy_targets)
    assert torch.equal(torch.nan_to_num(c.soft_value_targets, nan=9.0), torch.nan_to_num(b.soft_value_targets, nan=9.0))
    assert torch.equal(torch.nan_to_num(a.soft_value_targets, nan=9.0), torch.nan_to_num(b.soft_value_targets, nan=9.0))
    assert pops[0].games_finished == pops[1].games_finished == pops[2].games_finished > 0
    assert torch.equal(pops[0].outcome, pops[1].outcome) and torch.equal(pops[2].outcome, pops[1].outcome)
    d = pops[3].buffer.build()
    assert torch.equal(d.state_tensors, b.state_tensors) and torch.equal(d.policy_targets, b.policy_targets)
    assert torch.equal(torch.nan_to_num(d.value_targets, nan=9.0), torch.nan_to_num(b.value_targets, nan=9.0))
    assert pops[3].games_finished == pops[1].games_finished and pops[3].leaf_evals == pops[1].leaf_evals
    for q in (pops[0], pops[2], pops[3]):
        q.tail.check_overflow()
        for t in ("plies", "step_counts"):
            assert torch.equal(getattr(q, t), getattr(pops[1], t)), t
        assert torch.equal(q.states.board, pops[1].states.board)
    assert pops[0].leaf_evals == pops[1].leaf_evals


def test_root_self_play_runner_fused_equals_operator_chain():
    """self_play_v1_gpu: whole waves through the fused search give the same trajectories as the operator chain."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0"))
    kw = dict(num_games=96, mcts_simulations=16, temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
              exploration_weight=1.0, device="cuda:0", add_dirichlet_noise=False, sample_moves=False, max_game_plies=512,
              concurrent_games=48)
    a, sa = self_play_v1_gpu(net, fused_search=True, **kw)
    b, sb = self_play_v1_gpu(net, fused_search=False, **kw)
    assert sa.mcts_counters.get("fused_root_search") == 1 and "fused_root_search" not in sb.mcts_counters
    assert a.num_samples == b.num_samples > 96 * 100
    for k in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert (sa.black_wins, sa.white_wins, sa.draws) == (sb.black_wins, sb.white_wins, sb.draws)
    # the fused path also records, moves and finalises on the device (wave_tail.WaveTail): same bookkeeping
    assert sa.piece_delta_buckets == sb.piece_delta_buckets and sum(sa.piece_delta_buckets.values()) == 96
    assert sa.avg_game_length == sb.avg_game_length and sa.num_positions == sb.num_positions


def test_root_self_play_runner_continuous_waves_start_next_games_in_finished_slots():
    """80 sampled games through 32 slots: finished slots start the remaining games at once (lz_wave_reseat); every game
    is finalised exactly once, the bookkeeping adds up and the run is reproducible from the torch seed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0"))
    kw = dict(num_games=80, mcts_simulations=16, temperature_init=1.0, temperature_final=0.5, temperature_threshold=10,
              exploration_weight=1.0, device="cuda:0", add_dirichlet_noise=True, sample_moves=True, max_game_plies=160,
              concurrent_games=32)
    runs = []
    for _ in range(2):
        torch.manual_seed(77)
        runs.append(self_play_v1_gpu(net, **kw))
    (a, sa), (b, sb) = runs
    assert sa.mcts_counters.get("fused_root_search") == 1
    assert sa.num_games == 80 and sa.black_wins + sa.white_wins + sa.draws == 80
    assert sum(sa.piece_delta_buckets.values()) == 80
    assert a.num_samples == sa.num_positions == round(sa.avg_game_length * 80)
    assert not torch.isnan(a.value_targets).any() and not torch.isnan(a.soft_value_targets).any()
    assert a.value_targets.abs().max() <= 1.0 and (a.policy_targets.sum(dim=1) - 1.0).abs().max() < 1e-4
    assert bool((a.policy_targets[~a.legal_masks] == 0).all())
    # sampled games differ in length, so slots re-seat at different plies: more plies than one game, fewer than 3 waves
    assert a.num_samples > 80 * 20
    for k in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert sa.piece_delta_buckets == sb.piece_delta_buckets
    # sequential waves (the reference's loop) play the same number of games to the same kind of samples
    torch.manual_seed(77)
    c, sc = self_play_v1_gpu(net, continuous_waves=False, **{**kw, "num_games": 64})
    assert sc.black_wins + sc.white_wins + sc.draws == 64 and not torch.isnan(c.value_targets).any()


def test_tree_self_play_runner_with_legacy_waves():
    """self_play_tree_gpu(batch_k=8): the legacy search's waves inside the product runner -- same tensor contract,
    every game played to the ply cap, every search spends its full budget (root visits = sims)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import self_play_tree_gpu, PortableTreeMCTS
    from liuzhou_amd.mcts_gpu import GpuStateBatch
    torch.manual_seed(2)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0"))
    batch, stats = self_play_tree_gpu(net, num_games=70, mcts_simulations=20, temperature_init=1.0,
                                      temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                      device="cuda:0", max_game_plies=26, concurrent_games=35, batch_k=8)
    assert batch.num_samples == 70 * 26 == stats.num_positions
    assert torch.allclose(batch.policy_targets.sum(1), torch.ones(batch.num_samples, device="cuda:0"), atol=1e-4)
    assert bool((batch.policy_targets[~batch.legal_masks] == 0).all()) and bool(torch.isfinite(batch.value_targets).all())
    assert stats.mcts_counters["leaf_eval_count"] > 70 * 26 * 15
    # one search on the empty boards: every root ends with exactly `sims` visits over its children
    mcts = PortableTreeMCTS(net, 64, 50, "cuda:0", add_dirichlet_noise=False, sample_moves=False, batch_k=16)
    st = GpuStateBatch.initial(torch.device("cuda:0"), 64)
    mcts.search_batch(st, temperatures=torch.ones(64, device="cuda:0"))
    assert bool((mcts.engine.child_visits.sum(dim=1) == 50).all())


def _check_tree_trace(batch, stats, z, tag):
    """The reference's portable runner emits a game's samples when the game ends (game-major rows); the device arena
    records ply by ply (the order of the reference's v1 GPU runner).  Every game of the fixtures runs to the ply cap,
    so row (ply, game) of ours is row (game, ply) of the recording."""
    n = z[f"{tag}_policy_targets"].shape[0]
    games, _sims, plies = (int(x) for x in z[f"{tag}_config"])
    assert batch.num_samples == n == games * plies and stats.avg_game_length == float(plies)
    order = torch.arange(n).view(plies, games).t().reshape(-1)              # game-major view of ply-major rows
    g = lambda t: t.cpu()[order].numpy()
    want_states = np.unpackbits(z[f"{tag}_state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(g(batch.state_tensors), want_states)
    assert np.array_equal(g(batch.legal_masks), np.unpackbits(z[f"{tag}_legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(g(batch.policy_targets), z[f"{tag}_policy_targets"], atol=1e-6, rtol=0)
    np.testing.assert_array_equal(g(batch.value_targets), z[f"{tag}_value_targets"])
    np.testing.assert_allclose(g(batch.soft_value_targets), z[f"{tag}_soft_value_targets"], atol=1e-6, rtol=0)
    assert [stats.black_wins, stats.white_wins, stats.draws] == [int(x) for x in z[f"{tag}_outcome"]]


@pytest.mark.parametrize("tag,extra", [("a", {}), ("b", dict(policy_target_temperature=1.0,
                                                              policy_target_prior_pseudocount=0.5))])
def test_tree_runner_reproduces_reference_portable_selfplay_trace(tag, extra):
    """g10: the reference's own portable full-tree runner (v1/python/portable_self_play.py, tiny net seed 7, subtree
    reuse on every move, deterministic picks) recorded on CPU; `self_play_tree_gpu` -- device trees, device tail, the
    split-phase protocol with an external evaluator -- must produce the same samples.  The tiny random net gives
    near-uniform priors, and the recorded games turn on last-bit differences of the priors (a 1-ulp perturbation changes
    the trace at sample 12; a host CPU with another vector ISA already rounds the convolutions differently), so the
    evaluator replays the network outputs the reference run itself produced, recorded at its own hand-off
    (`PortableMCTS.evaluate_states` -> `PriorEvaluator`; oracle/gen_golden.py)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.tree_engine import PriorEvaluator, self_play_tree_gpu
    from oracle import lz_oracle as O
    from oracle import selfplay_oracle as SO
    from tests.tree_parity import unpack_packed
    z = load("g10_tree_selfplay.npz")
    games, sims, max_plies = (int(x) for x in z[f"{tag}_config"])
    evaluate = SO.make_table_evaluator(z[f"{tag}_eval_planes"], z[f"{tag}_eval_priors"], z[f"{tag}_eval_values"])

    def fn(planes, packed):
        pri, val = evaluate(unpack_packed(packed.cpu().numpy()))
        return torch.from_numpy(pri).to(planes.device), torch.from_numpy(val).to(planes.device)

    for tail in (True, False):
        batch, stats = self_play_tree_gpu(PriorEvaluator(fn), num_games=games, mcts_simulations=sims, temperature_init=1.0,
                                          temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                          device="cuda:0", add_dirichlet_noise=False, soft_value_k=2.0,
                                          opening_random_moves=0, max_game_plies=max_plies, sample_moves=False,
                                          concurrent_games=games, reuse_tree=True, device_tail=tail,
                                          collect_timing=tail, **extra)
        _check_tree_trace(batch, stats, z, tag)
        if tail:      # timing buckets of the reference's runner are filled (self_play_gpu_runner.py:276-281)
            assert stats.step_timing_ms["root_puct_ms"] > 0 and stats.step_timing_calls["root_puct_ms"] > 0
            assert stats.step_timing_ms["self_play_step_ms"] > 0 and stats.step_timing_ms["finalize_ms"] > 0
            assert abs(sum(stats.step_timing_ratio.values()) - 1.0) < 1e-6


def test_tree_runner_with_a_module_as_external_evaluator():
    """Any module with ChessNet.forward's outputs can drive the device trees (heads -> priors formed by the expand
    kernel): the tiny net, for which no fused kernel exists, plays whole games; samples are well-formed and the first
    plies -- before last-bit differences of the softmax can matter -- equal the reference trace."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.tree_engine import self_play_tree_gpu
    z = load("g10_tree_selfplay.npz")
    games, sims, max_plies = (int(x) for x in z["a_config"])
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to("cuda:0")
    batch, stats = self_play_tree_gpu(model, num_games=games, mcts_simulations=sims, temperature_init=1.0,
                                      temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                      device="cuda:0", add_dirichlet_noise=False, sample_moves=False,
                                      max_game_plies=max_plies, concurrent_games=games, evaluator="module")
    assert batch.num_samples == games * max_plies
    n = z["a_policy_targets"].shape[0]
    want = np.unpackbits(z["a_state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(batch.state_tensors[0].cpu().numpy(), want[0])      # the opening position
    assert torch.allclose(batch.policy_targets.sum(1), torch.ones(batch.num_samples, device="cuda:0"), atol=1e-5)
    assert bool((batch.policy_targets[~batch.legal_masks] == 0).all())


def test_fused_root_search_rng_is_independent_of_the_batch_split():
    """Root noise and sampled picks of the fused root search come from the per-game counter RNG: one search over all
    games == two half-size searches on two streams (DualStreamRootSearch), noise and sampling on, over several plies;
    another seed plays other moves."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.mcts_gpu import GpuStateBatch
    from liuzhou_amd.root_search_fused import DualStreamRootSearch, FusedRootSearch
    dev = "cuda:0"
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    B = 70
    kw = dict(add_dirichlet_noise=True, sample_moves=True, seed=99)
    one, two = FusedRootSearch(net, B, 32, dev, **kw), DualStreamRootSearch(net, B, 32, dev, **kw)
    other = FusedRootSearch(net, B, 32, dev, **{**kw, "seed": 100})
    s1, s2 = GpuStateBatch.initial(dev, B), GpuStateBatch.initial(dev, B)
    temps = torch.ones(B, device=dev)
    differs = False
    for ply in range(6):
        plies = torch.full((B,), ply, dtype=torch.int64, device=dev)
        o1 = one.search_batch(s1, temperatures=temps, rng_plies=plies)
        o2 = two.search_batch(s2, temperatures=temps, rng_plies=plies)
        o3 = other.search_batch(s1, temperatures=temps, rng_plies=plies)
        assert torch.equal(o1.chosen_action_indices, o2.chosen_action_indices), ply
        assert torch.equal(o1.policy_dense, o2.policy_dense), ply
        differs = differs or not torch.equal(o1.chosen_action_indices, o3.chosen_action_indices)
        for s, o in ((s1, o1), (s2, o2)):
            p = torch.zeros(B, dtype=torch.int64, device=dev); d = torch.zeros(B, dtype=torch.bool, device=dev)
            v0_core.self_play_step_inplace(*s.tensors(), p, d, torch.arange(B, device=dev), o.chosen_action_codes.clone(),
                                           o.terminal_mask.clone(), o.chosen_valid_mask.clone(), 512, 2.0)
    assert differs


def test_tree_runner_switches_to_compact_lists_in_the_drain_and_plays_the_same_games(monkeypatch):
    """The product default of an engine whose full launch is several network passes per CU (10x128, 4 096 slots = 2
    passes): dense launches while nearly all games are live, compact evaluation lists once the wave has drained by a pass
    per CU (WaveTail.live_estimate, two plies old).  One full-length wave played three ways -- dense only, lists always,
    automatic -- gives byte-identical trajectories."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import clear_engine_cache, self_play_tree_gpu
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b10c128"]).eval().to("cuda:0"))
    out = {}
    for mode in ("0", "1", "auto"):
        if mode == "auto":
            monkeypatch.delenv("LZ_TREE_COMPACT", raising=False)
        else:
            monkeypatch.setenv("LZ_TREE_COMPACT", mode)
        batch, st = self_play_tree_gpu(net, num_games=4096, mcts_simulations=6, temperature_init=1.0, temperature_final=0.1,
                                       temperature_threshold=10, exploration_weight=1.0, device="cuda:0",
                                       add_dirichlet_noise=True, sample_moves=True, concurrent_games=4096,
                                       max_game_plies=512, seed=77)
        out[mode] = (batch, st)
        clear_engine_cache()
    ref = out["0"][0]
    for mode in ("1", "auto"):
        b = out[mode][0]
        for f in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"):
            assert torch.equal(getattr(b, f), getattr(ref, f)), (mode, f)
    c0, c1, ca = (out[m][1].mcts_counters for m in ("0", "1", "auto"))
    plies = ca["plies_launched"]
    assert c0["list_searches"] == 0 and c0["compact_eval_lists"] == 0
    assert c1["list_searches"] >= plies - 1 and ca["compact_eval_lists"] == 1
    # (with 6 simulations of a random-init net every game runs into the 144-move limit within a ply or two of the others:
    #  the wave ends too abruptly for the two-plies-old estimate to see it half empty -- the switch itself is exercised
    #  move by move in test_gpu_tree.py::test_alternating_launch_forms..., and by the full-length C3 run of
    #  profiles/r05_c3_full_length.json, where the lists carry the second half of the run)
    assert ca["list_searches"] <= plies
    assert c1["leaf_eval_count"] <= ca["leaf_eval_count"] <= c0["leaf_eval_count"]
    print(f"{plies} plies: lists in {ca['list_searches']} of them; launched evaluations dense {c0['leaf_eval_count']}, "
          f"auto {ca['leaf_eval_count']}, lists always {c1['leaf_eval_count']}")
