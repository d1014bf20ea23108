"""GPU: BASELINE.json's single-GPU configurations at FULL size, through size-independent properties.

The oracle cannot follow 4 096 x 200 or 16 384 x 800 searches in seconds, so these runs -- the exact objects bench.py
times (`SteadyStateTreeSelfPlay`: two streams for C2, one captured search per move, subtree reuse, Philox noise, sampled
moves, device tail) -- are checked through invariants every correct run satisfies (reference semantics:
v1/python/portable_mcts.py:123-138,418-506 visit accounting; self_play_gpu_runner.py:205-247 trajectory rows):

  * every non-terminal root ends a move with exactly `sims` new visits: sum of its children's visits == root visits for a
    fresh root, == root visits - 1 for a kept root (its own expansion visit), and root visits >= sims;
  * no overflow counter fired, every row fits the exact 360-byte record of the C4 gather (not-representable counter 0, round
    trip bit-exact); dropped / pruned subtrees are reported;
  * trajectory rows are well formed: model input == the operator's encoding of the state that was searched, legal mask ==
    lz_encode_actions_fast of it, the policy target is a distribution on the legal set;
  * the same seed plays the same moves again (bit-identical picks and policies).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engines(pop):
    m = pop.mcts
    return [p.engine for p in getattr(m, "parts", [])] or [m.engine]


def _population(name, games, sims, steps, seed, reuse_factor=-1.0):
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay
    dev = torch.device(DEV)
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
    torch.manual_seed(seed)
    pop = SteadyStateTreeSelfPlay(net, games, sims=sims, device=dev, seed=seed, reuse_tree=True, reuse_factor=reuse_factor,
                                  dual_stream=True, arena_rows=games * (steps + 4))
    pop.preroll(120)
    pop.prepare()
    return pop


def _checked_step(pop, sims, first_move):
    """One step of the harness with every invariant checked; returns (chosen indices, policy rows) of the step."""
    from liuzhou_amd import v0_core
    from liuzhou_amd.mcts_gpu import states_to_model_input
    p = pop.pop
    dev = p.dev
    B = pop.B
    before = p.states._map(lambda t: t.clone())
    row0 = p.buffer.sync_cursor()
    pop.step()
    torch.cuda.synchronize(dev)
    row1 = p.buffer.sync_cursor()
    assert row1 - row0 == B, "every slot of the steady-state population records one row per step"
    # ---- visit accounting, engine by engine ----
    chosen, start = [], 0
    for e in _engines(pop):
        term = e.terminal_mask.cpu().numpy().astype(bool)
        cnt = e.child_count.cpu().numpy()
        vis = e.child_visits.cpu().numpy()
        rv = e.buf["root_visits"].cpu().numpy()
        live = ~term
        assert live.any()
        assert (cnt[live] >= 1).all() and (cnt[live] <= 72).all()
        csum = np.array([vis[g, :cnt[g]].sum() for g in range(e.B)])
        assert (rv[live] >= sims).all()
        fresh = live & (rv == sims)
        kept = live & (rv > sims)
        assert (csum[fresh] == sims).all(), "a fresh root's children hold exactly `sims` visits"
        assert (csum[kept] == rv[kept] - 1).all(), "a kept root's children hold its visits minus its own expansion visit"
        if first_move:
            assert not kept.any()
        pick = e.chosen_index.cpu().numpy()
        act = e.child_action.cpu().numpy()
        for g in np.flatnonzero(live)[:: max(1, e.B // 257)]:
            assert pick[g] in act[g, :cnt[g]], "the played move is one of the root's children"
        assert (pick[term] == -1).all()
        chosen.append(pick)
        start += e.B
    assert start == B
    # ---- trajectory rows of this step ----
    a_state, a_legal, a_policy, a_value, a_soft, a_sign = p.buffer.arena()
    rows = slice(row0, row1)
    want_mask, _ = v0_core.encode_actions_fast(*before.tensors()[:10], 36, 144, 36, 4)
    assert torch.equal(a_legal[rows], want_mask), "legal mask row != lz_encode_actions_fast of the searched state"
    assert torch.equal(a_state[rows], states_to_model_input(before)), "model-input row != encoding of the searched state"
    pol = a_policy[rows]
    assert torch.isfinite(pol).all() and (pol >= 0).all()
    assert float((pol * (~want_mask)).abs().max()) == 0.0, "policy mass outside the legal set"
    has_legal = want_mask.any(dim=1)
    s = pol.sum(dim=1)
    assert float((s[has_legal] - 1.0).abs().max()) <= 1e-5
    assert torch.equal(a_sign[rows].to(torch.int64), before.current_player.to(torch.int64))
    if p.tail is not None:
        p.tail.check_overflow()
    # the wire format of the C4 gather holds every row of the step exactly (`not representable` counter == 0)
    from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
    from liuzhou_amd.trajectory_codec import pack_batch, unpack_records
    mine = TensorSelfPlayBatch(a_state[rows], a_legal[rows], a_policy[rows], a_value[rows], a_soft[rows])
    rec, bad = pack_batch(mine, return_bad=True)
    assert int(bad.item()) == 0, "rows the 360-byte record cannot represent"
    back = unpack_records(rec)
    assert torch.equal(back.state_tensors, mine.state_tensors) and torch.equal(back.legal_masks, mine.legal_masks)
    assert torch.equal(back.policy_targets, mine.policy_targets)
    return np.concatenate(chosen), pol.clone()


def _dropped(pop):
    return [sum(int(e.reuse_dropped[k].item()) for e in _engines(pop)) for k in (0, 1)]


def _pool_ok(pop, max_tree_gbytes):
    """The engines' edge pools: no expansion was refused, the peak use leaves head-room, every chunk is either free or on
    exactly one game's list, and the tree memory is what DESIGN.md section 4 says."""
    total = 0
    for e in _engines(pop):
        st = e.pool_status()
        assert st["refused_expansions"] == 0, st
        assert st["fewest_free"] > 0.4 * st["chunks"], f"edge pool nearly exhausted: {st}"
        owned = int(e.buf["n_chunks"].sum(dtype=torch.int64).item())
        assert owned + st["free"] == st["chunks"], "a chunk was lost or handed out twice"
        assert int(e.buf["n_chunks"].max().item()) <= e.chunk_cap
        total += e.hbm_bytes()
    assert total <= max_tree_gbytes * 2**30, f"tree memory {total / 2**30:.1f} GiB"
    return total / 2**30


def _advance_launch_us(fn):
    """Average duration of the `tree_advance_kernel` launches inside `fn()` (HIP events around the launch,
    lz_prof_aux_summary kind 1) -- the one kernel whose cost grows with the arenas, bounded by its slowest wave."""
    import ctypes
    from liuzhou_amd import _lib as LZ
    LZ.check(LZ.lib().lz_prof_enable(1), "prof_enable")
    try:
        out = fn()
        torch.cuda.synchronize()
        ms, n, units = ctypes.c_double(0.0), ctypes.c_int64(0), ctypes.c_int64(0)
        LZ.check(LZ.lib().lz_prof_aux_summary(1, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(units)), "prof_aux_summary")
    finally:
        LZ.lib().lz_prof_enable(0)
    assert n.value >= 1, "no tree_advance_kernel launch was bracketed"
    return out, ms.value / n.value * 1e3


def test_c2_full_size_three_moves_and_same_seed_same_games():
    """C2 = 4 096 concurrent games, 200 simulations per move, 6x64 net, two streams: three consecutive moves with kept
    subtrees, checked move by move; a second population with the same seed plays bit-identical moves."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    runs = []
    for _ in range(2):
        pop = _population("b6c64", 4096, 200, 3, seed=9973, reuse_factor=8.0)
        assert pop.dual_stream and all(e.B == 2048 for e in _engines(pop))
        steps = [_checked_step(pop, 200, first_move=True), _checked_step(pop, 200, first_move=False)]
        # (the timing guard on the subtree compaction lives in tests/test_perf_guards.py, marker `perf`: a throttled or
        #  shared box must not turn a parity run red)
        last, adv_us = _advance_launch_us(lambda: _checked_step(pop, 200, first_move=False))
        steps.append(last)
        print(f"C2 full size: tree_advance_kernel {adv_us:.0f} us per half-batch launch")
        kept_any = any(int((e.buf["root_visits"] > 200).sum()) > 0 for e in _engines(pop))
        assert kept_any, "no game kept a subtree over three moves"
        assert _dropped(pop) == [0, 0]
        print(f"C2 full size: [dropped, pruned] subtrees {_dropped(pop)}, tree memory {_pool_ok(pop, 8.0):.1f} GiB")
        runs.append(steps)
        del pop
        torch.cuda.empty_cache()
    for (ca, pa), (cb, pb) in zip(*runs):
        assert np.array_equal(ca, cb), "same seed, different picks"
        assert torch.equal(pa, pb), "same seed, different policy targets"


def test_c3_full_size_one_move_and_one_continued_move():
    """C3 = 16 384 concurrent games, 800 simulations per move, 10x128 net, the whole search of a move as one hipGraph:
    one fresh move and one move that continues the kept subtrees, arenas sized from free memory as in bench.py."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gc
    gc.collect()
    torch.cuda.empty_cache()                                    # give back what earlier tests' allocations left cached
    free, _ = torch.cuda.mem_get_info(torch.device(DEV))
    if free < 150 * (1 << 30):
        pytest.skip("C3's tree arenas need most of a 288 GB device")
    pop = _population("b10c128", 16384, 800, 2, seed=9973)
    assert not pop.dual_stream and pop.mcts.use_graph
    _checked_step(pop, 800, first_move=True)
    _, adv_us = _advance_launch_us(lambda: _checked_step(pop, 800, first_move=False))
    e = _engines(pop)[0]
    assert int((e.buf["root_visits"] > 800).sum()) > 0, "no game continued a kept subtree"
    assert not pop.mcts.graph_retry_off
    assert _dropped(pop) == [0, 0]
    # round 4: 24 GB of node arenas + 38.5 GB of edge pool (rounds 1-3: ~210 GB of worst-case regions per game)
    print(f"C3 full size: tree_advance_kernel {adv_us:.0f} us; arena factor {e.reuse_factor}, [dropped, pruned] subtrees {_dropped(pop)}, "
          f"tree memory {_pool_ok(pop, 120.0):.1f} GiB")
    del pop, e
    gc.collect()
    torch.cuda.empty_cache()
