"""GPU: device-resident full-tree PUCT engine vs the oracle and the reference's recorded searches."""
import numpy as np
import pytest
import torch

from oracle import lz_oracle as O
from tests.golden_utils import load, states, FIELDS
from tests.tree_parity import run_injected_parity, unpack_packed, to_gpu_batch, engine_visits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_pack_roundtrip_and_planes():
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    z = load("g1_rules.npz")
    st = states(z, "s")
    B = st["board"].shape[0]
    eng = TreeEngine(B, 2, DEV)
    eng.set_roots(to_gpu_batch(st, DEV))
    back = unpack_packed(eng.buf["root_state"].cpu().numpy())
    for f in FIELDS:
        assert np.array_equal(np.asarray(back[f]).reshape(B, -1).astype(np.int64),
                              np.asarray(st[f]).reshape(B, -1).astype(np.int64)), f
    eng.begin()
    planes = eng.leaf_planes().cpu().numpy()
    assert np.array_equal(planes, O.states_to_model_input(st))


@pytest.mark.parametrize("sims,games,seed", [(8, 256, 1), (64, 256, 2), (200, 128, 3)])
def test_tree_visit_counts_bit_exact_vs_oracle(sims, games, seed):
    """Visit counts / priors / picks bit-exact on fixed seeds with an injected evaluator (batch of mixed phases)."""
    _need_gpu()
    run_injected_parity(DEV, num_games=games, sims=sims, seed=seed)


def test_tree_with_root_noise_and_temperature():
    _need_gpu()
    rng = np.random.default_rng(11)
    noise = rng.gamma(0.3, 1.0, size=(96, 80)).astype(np.float32) + 1e-6
    run_injected_parity(DEV, num_games=96, sims=48, seed=4, noise=noise, eps=0.25, temperature=0.1, c=1.5)


def test_tree_edge_states_terminal_and_no_legal():
    """Representative + terminal states of the reference tests (root terminal, no-legal MARK_SELECTION, ...)."""
    _need_gpu()
    z = load("g2_edges.npz")
    st = states(z, "s")
    run_injected_parity(DEV, sims=32, states={f: np.asarray(st[f]) for f in FIELDS})


def test_tree_matches_reference_portable_mcts_recorded_searches():
    """tests/golden/g5_tree.npz: PortableMCTS (and src/mcts.py batch_K=1) visit counts with the reference's own
    network outputs replayed as the evaluator."""
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    z = load("g5_tree.npz")
    roots = states(z, "r")
    est = states(z, "e")
    case_sims = z["case_sims"]; case_noise = z["case_noise_flag"]
    for sims in sorted(set(case_sims.tolist())):
        for nflag in (False, True):
            cases = np.nonzero((case_sims == sims) & (case_noise == nflag))[0]
            if cases.size == 0:
                continue
            B = cases.size
            rst = {f: np.ascontiguousarray(np.asarray(roots[f])[z["case_root"][cases]]) for f in FIELDS}
            eng = TreeEngine(B, int(sims), DEV, 1.0)
            eng.set_roots(to_gpu_batch(rst, DEV))
            eng.begin()
            cursor = z["case_eval_start"][cases].copy()
            noise = torch.from_numpy(z["case_noise"][cases].astype(np.float32)).to(DEV) if nflag else None

            def complete(is_root):
                kind = eng.buf["leaf_kind"].cpu().numpy()
                leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
                pri = np.zeros((B, 220), np.float32); val = np.zeros(B, np.float32)
                for b in range(B):
                    if kind[b] != 1:
                        continue
                    k = int(cursor[b]); cursor[b] += 1
                    for f in FIELDS:   # the engine asks for exactly the state the reference evaluated
                        assert np.array_equal(np.asarray(leaf[f])[b].reshape(-1).astype(np.int64),
                                              np.asarray(est[f])[k].reshape(-1).astype(np.int64)), (sims, b, f)
                    pri[b] = z["eval_priors"][k]; val[b] = z["eval_value"][k]
                eng.expand(is_root=is_root, values=torch.from_numpy(val).to(DEV), priors220=torch.from_numpy(pri).to(DEV),
                           noise=noise if is_root else None, epsilon=0.25)

            complete(True)
            for _ in range(int(sims)):
                eng.select()
                complete(False)
            assert np.array_equal(cursor, z["case_eval_start"][cases] + z["case_eval_count"][cases])
            eng.finish(torch.ones(B, device=DEV), None)
            got_v, got_p = engine_visits(eng)
            assert np.array_equal(got_v, z["case_visits"][cases]), (sims, nflag)
            np.testing.assert_allclose(got_p, z["case_root_priors"][cases], atol=1e-6, rtol=0)
            np.testing.assert_allclose(eng.policy_dense.cpu().numpy(), z["case_policy_t1"][cases], atol=1e-6, rtol=0)
            np.testing.assert_allclose(eng.root_value.cpu().numpy(), z["case_root_value"][cases], atol=1e-6, rtol=0)
            assert np.array_equal(eng.chosen_index.cpu().numpy(), z["case_chosen"][cases])
            eng.finish(torch.full((B,), 0.1, device=DEV), None)
            np.testing.assert_allclose(eng.policy_dense.cpu().numpy(), z["case_policy_t01"][cases], atol=1e-6, rtol=0)


def test_fused_search_runs_and_conserves_visits():
    """Whole search enqueued from C++ with the fused network in the loop: sum of child visits == sims."""
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import PortableTreeMCTS
    from liuzhou_amd.mcts_gpu import GpuStateBatch
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    z = load("g1_rules.npz")
    st = states(z, "s")
    idx = np.random.default_rng(0).integers(0, st["board"].shape[0], 512)
    batch = to_gpu_batch({f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}, DEV)
    mcts = PortableTreeMCTS(net, 512, 50, DEV)
    out = mcts.search_batch(batch, temperatures=torch.ones(512, device=DEV))
    vis = mcts.engine.child_visits.cpu().numpy(); cnt = mcts.engine.child_count.cpu().numpy()
    for g in range(512):
        assert int(vis[g, :cnt[g]].sum()) == 50
    pol = out.policy_dense
    assert torch.allclose(pol.sum(1), torch.ones(512, device=DEV), atol=1e-5)
    assert bool((pol[~out.legal_mask] == 0).all())
    assert bool(out.chosen_valid_mask.all())
    picked = out.legal_mask.gather(1, out.chosen_action_indices.view(-1, 1)).view(-1)
    assert bool(picked.all())


@pytest.mark.parametrize("sims,games,moves,noise,chunk", [(48, 48, 4, True, None), (200, 24, 3, False, None),
                                                           (160, 32, 5, True, 128), (96, 40, 4, True, 256)])
def test_tree_reuse_bit_exact_vs_oracle_over_consecutive_moves(sims, games, moves, noise, chunk):
    """a21 (advance_root): the played child's subtree is kept, compacted in place, re-noised and searched on.  With chunks
    of 128 / 256 records the edge pool hands out a new chunk every few expansions and every compaction re-packs the kept runs
    across many chunk boundaries: still the oracle's visit counts bit for bit, and no chunk is lost."""
    _need_gpu()
    from tests.tree_parity import run_injected_reuse_parity
    eng, kept = run_injected_reuse_parity(DEV, num_games=games, sims=sims, moves=moves, seed=sims, with_noise=noise,
                                          edge_chunk=chunk)
    assert kept > 0, "no game ever kept a subtree -- the test did not exercise the reuse path"
    st = eng.pool_status()
    assert st["refused_expansions"] == 0 and int(eng.buf["n_chunks"].sum()) + st["free"] == st["chunks"]
    if chunk is not None:
        assert int(eng.buf["n_chunks"].max()) >= 4, "the small-chunk case did not cross chunk boundaries"


def test_tree_reuse_falls_back_to_fresh_roots():
    """reset flags, unknown played actions and states that are not the child all start a fresh tree; a subtree that
    would not leave room for the next search is dropped and counted."""
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    from tests.tree_parity import hash_evaluator, unpack_packed
    from oracle import lz_oracle as O
    z = load("g1_rules.npz")
    st = states(z, "s")
    idx = np.random.default_rng(5).integers(0, st["board"].shape[0], 32)
    sub = {f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}
    B, sims = 32, 40

    def search(eng):
        for s in range(sims + 1):
            if s:
                eng.select()
            leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
            pri, val = hash_evaluator(leaf)
            eng.expand(is_root=(s == 0), values=torch.from_numpy(val).to(DEV), priors220=torch.from_numpy(pri).to(DEV))
        eng.finish(torch.full((B,), 0.1, device=DEV), None)

    for factor, expect_drop in ((4.0, False), (0.02, True)):
        eng = TreeEngine(B, sims, DEV, 1.0, reuse_factor=factor)
        eng.set_roots(to_gpu_batch(sub, DEV)); eng.begin(); search(eng)
        chosen = eng.chosen_index.cpu().numpy()
        term = eng.terminal_mask.cpu().numpy()
        nxt = [O.state_from_batch(sub, i) if term[i] else O.apply_index(O.state_from_batch(sub, i), int(chosen[i]))
               for i in range(B)]
        nb = O.batch_from_states(nxt)
        reset = torch.zeros(B, dtype=torch.uint8, device=DEV); reset[::4] = 1
        played = eng.chosen_index.clone(); played[1::4] = -1
        wrong = {f: np.array(nb[f]) for f in FIELDS}
        for i in range(2, B, 4):                       # a state that is not the played child's
            for f in FIELDS:
                wrong[f][i] = np.asarray(sub[f])[i]
        eng.set_roots(to_gpu_batch(wrong, DEV))
        eng.advance(played, reset)
        kind = eng.buf["leaf_kind"].cpu().numpy()
        fresh_expected = np.zeros(B, bool); fresh_expected[::4] = True; fresh_expected[1::4] = True; fresh_expected[2::4] = True
        assert not np.any(kind[fresh_expected] == 3)
        rest = ~fresh_expected & ~term
        dropped = int(eng.reuse_dropped[0].item())
        if expect_drop:
            assert dropped > 0 and not np.any(kind == 3)
        else:
            assert dropped == 0 and np.any(kind[rest] == 3)
        # whatever was kept or dropped, the next search is a valid one
        search(eng)
        vis = eng.child_visits.cpu().numpy(); cnt = eng.child_count.cpu().numpy()
        for g in np.nonzero(kind == 1)[0]:
            assert int(vis[g, :cnt[g]].sum()) == sims


def test_advance_prunes_a_subtree_that_does_not_fit_instead_of_dropping_it():
    """The reference's tree is unbounded; ours lives in a per-game arena.  A kept subtree that would leave no room for the
    next search is cut to its oldest part (expansion order): the arena stays structurally sound, the new root's own
    statistics are exactly the played child's, `pruned` counts the games, nothing is dropped whole, and the next search
    adds exactly `sims` visits."""
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    from tests.tree_parity import hash_evaluator, unpack_packed, game_tree
    from oracle import lz_oracle as O
    z = load("g1_rules.npz")
    st = states(z, "s")
    idx = np.random.default_rng(11).integers(0, st["board"].shape[0], 24)
    cur = {f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}
    B, sims = 24, 320
    # room for 96 kept nodes: most kept subtrees are larger; chunks of 128 edges, so that every few expansions take a
    # new chunk from the pool and the compaction of a kept subtree crosses many chunk boundaries
    eng = TreeEngine(B, sims, DEV, 1.0, reuse_factor=0.3, edge_chunk=128)
    node_budget = eng.node_cap - (sims + 1)

    def search(first):
        for s in range(sims + 1):
            if s:
                eng.select()
            leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
            pri, val = hash_evaluator(leaf)
            eng.expand(is_root=(s == 0), values=torch.from_numpy(val).to(DEV), priors220=torch.from_numpy(pri).to(DEV))
        eng.finish(torch.full((B,), 0.1, device=DEV), None)

    def arena(g):
        return game_tree(eng, g)

    def chunks_consistent():
        """Every chunk is either free or owned by exactly one game; every run lies inside one chunk its game owns."""
        nc = eng.buf["n_chunks"].cpu().numpy()
        lists = eng.buf["chunk_list"].view(B, eng.chunk_cap).cpu().numpy()
        top = int(eng.buf["pool_top"].item())
        free = eng.buf["free_chunks"].cpu().numpy()[:top].tolist()
        owned = [c for g in range(B) for c in lists[g, :nc[g]].tolist()]
        assert len(set(free)) == len(free) and len(set(owned)) == len(owned) and not (set(free) & set(owned))
        assert len(free) + len(owned) == eng.pool_chunks
        for g in range(B):
            mine = set(lists[g, :nc[g]].tolist())
            nodes, _runs = game_tree(eng, g)
            for nd in nodes:
                e0, n = int(nd["edge_begin"]), int(nd["nedges"])
                if n > 0:
                    assert e0 // eng.edge_chunk in mine and (e0 + n - 1) // eng.edge_chunk == e0 // eng.edge_chunk

    eng.set_roots(to_gpu_batch(cur, DEV)); eng.begin(); search(True)
    pruned_total = 0
    for move in range(3):
        chosen = eng.chosen_index.cpu().numpy()
        term = eng.terminal_mask.cpu().numpy()
        before = [arena(g) for g in range(B)]
        nxt = [O.state_from_batch(cur, i) if term[i] else O.apply_index(O.state_from_batch(cur, i), int(chosen[i]))
               for i in range(B)]
        cur = O.batch_from_states(nxt)
        eng.set_roots(to_gpu_batch(cur, DEV))
        eng.advance()
        kind = eng.buf["leaf_kind"].cpu().numpy()
        chunks_consistent()
        for g in range(B):
            nodes, runs = arena(g)
            if kind[g] != 3:
                continue
            assert nodes.size <= node_budget
            # structure: parents / child links consistent, ids ascend from parent to child, child runs where the edge says
            for i, nd in enumerate(nodes):
                n = int(nd["nedges"])
                assert n >= 1 and len(runs[i]) == n
                assert int(nd["parent"]) == (-1 if i == 0 else int(nd["parent"])) and int(nd["parent"]) < i
                for e in runs[i]:
                    c = int(e["child"])
                    if c >= 0:
                        assert i < c < nodes.size and int(nodes[c]["parent"]) == i
                        assert int(e["cbegin"]) == int(nodes[c]["edge_begin"]) and int(e["cn"]) == int(nodes[c]["nedges"])
            # the new root's own statistics are the played child's, bit for bit (action, N | info, W, P)
            onodes, oruns = before[g]
            ce = oruns[0][oruns[0]["act"] == chosen[g]][0]
            want = oruns[int(ce["child"])]
            got = runs[0]
            for f in ("act", "n_info", "W", "P"):
                assert got[f].tobytes() == want[f].tobytes(), (move, g, f)
            assert int(eng.buf["root_visits"][g]) == int(ce["n_info"] & 0xFFFFFF)
        dropped, pruned = eng.reuse_dropped.tolist()
        assert dropped == 0
        pruned_total = pruned
        rv0 = eng.buf["root_visits"].cpu().numpy().copy()
        search(False)
        rv1 = eng.buf["root_visits"].cpu().numpy()
        live = ~eng.terminal_mask.cpu().numpy()
        assert ((rv1 - rv0)[live & (kind == 3)] == sims).all()
        vis = eng.child_visits.cpu().numpy(); cnt = eng.child_count.cpu().numpy()
        for g in np.nonzero(live & (kind == 1))[0]:
            assert int(vis[g, :cnt[g]].sum()) == sims
    assert pruned_total > 0, "no subtree was pruned: the test did not exercise the cut"


def test_tree_finish_policy_target_options_and_uniform_openings():
    """policy_target_temperature / prior pseudocount (portable_mcts.py:150-205, :690-700) and the uniform pick of
    the opening plies (:709-712)."""
    _need_gpu()
    from oracle import lz_oracle as O
    eng = run_injected_parity(DEV, num_games=96, sims=40, seed=11, temperature=0.1)
    B = eng.B
    cnt = eng.child_count.cpu().numpy(); act = eng.child_action.cpu().numpy()
    vis = eng.child_visits.cpu().numpy(); pri = eng.child_prior.cpu().numpy()
    temps = torch.full((B,), 0.1, device=DEV)
    tt = torch.full((B,), 1.0, device=DEV)
    u = torch.rand(B, device=DEV)
    force = (torch.arange(B, device=DEV) % 2 == 0)
    before = eng.chosen_index.clone()
    eng.finish(temps, u, tt, 0.5, force, sample_moves=False)
    pol = eng.policy_dense.cpu().numpy(); chosen = eng.chosen_index.cpu().numpy(); uu = u.cpu().numpy()
    for g in range(B):
        k = int(cnt[g])
        if k == 0:
            continue
        want = np.zeros(220, np.float32)
        want[act[g, :k]] = O.policy_from_visits(vis[g, :k], 1.0, pri[g, :k], 0.5)
        np.testing.assert_allclose(pol[g], want, atol=2e-6, rtol=0)
        if g % 2 == 0:
            assert chosen[g] == act[g, min(k - 1, int(np.float32(uu[g]) * np.float32(k)))]
        else:
            assert chosen[g] == int(before[g])          # deterministic pick untouched by the target options


def test_fused_search_with_subtree_reuse_accumulates_visits():
    """PortableTreeMCTS(reuse_tree=True): second search of the same games continues from the kept subtree -- the
    root's children carry the visits they had as grandchildren plus the new simulations."""
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import PortableTreeMCTS
    from liuzhou_amd import v0_core
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    z = load("g1_rules.npz")
    st = states(z, "s")
    B, sims = 256, 60
    idx = np.random.default_rng(2).integers(0, st["board"].shape[0], B)
    batch = to_gpu_batch({f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}, DEV)
    for use_graph in (False, True):
        mcts = PortableTreeMCTS(net, B, sims, DEV, reuse_tree=True, sample_moves=False, use_graph=use_graph)
        cur = batch.select(torch.arange(B, device=DEV))
        plies = torch.zeros(B, dtype=torch.int64, device=DEV); done = torch.zeros(B, dtype=torch.bool, device=DEV)
        all_idx = torch.arange(B, device=DEV)
        prev_child_visits = None
        for mv in range(4):
            out = mcts.search_batch(cur, temperatures=torch.full((B,), 0.1, device=DEV), active=~done)
            e = mcts.engine
            vis = e.child_visits.cpu().numpy(); cnt = e.child_count.cpu().numpy(); act = e.child_action.cpu().numpy()
            kind_total = np.array([int(vis[g, :cnt[g]].sum()) for g in range(B)])
            live = (~done).cpu().numpy() & ~out.terminal_mask.cpu().numpy()
            if mv == 0:
                assert np.all(kind_total[live] == sims)
            else:
                # kept root: visits carried over = (visits of the played child - 1), then + sims
                carried = np.maximum(prev_child_visits - 1, 0)
                assert np.all(kind_total[live] == carried[live] + sims), (mv, use_graph)
            chosen = e.chosen_index.cpu().numpy()
            prev_child_visits = np.array([vis[g, list(act[g, :cnt[g]]).index(chosen[g])] if live[g] else 0
                                          for g in range(B)])
            v0_core.self_play_step_inplace(*cur.tensors(), plies, done, all_idx, out.chosen_action_codes,
                                           out.terminal_mask, out.chosen_valid_mask, 512, 2.0)
        assert mcts.engine.reuse_dropped.tolist() == [0, 0]
        assert prev_child_visits.max() > 1


def test_dual_stream_search_equals_single_engine_search():
    """Two half-size engines on two streams (tree kernel of one half overlapping the network of the other) give
    exactly the results of one engine over all games with the same network kernel configuration."""
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import DualStreamTreeMCTS, PortableTreeMCTS
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    z = load("g1_rules.npz")
    st = states(z, "s")
    B, sims = 257, 40                                            # odd: unequal halves
    idx = np.random.default_rng(4).integers(0, st["board"].shape[0], B)
    batch = to_gpu_batch({f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}, DEV)
    temps = torch.full((B,), 0.1, device=DEV)
    kw = dict(add_dirichlet_noise=False, sample_moves=False, reuse_tree=True)
    single = PortableTreeMCTS(net.variant(half_workgroups=True), B, sims, DEV, **kw)
    dual = DualStreamTreeMCTS(net, B, sims, DEV, **kw)
    for use_graph in (False, True):
        single.use_graph = dual.use_graph = use_graph
        single.reset_trees(); dual.reset_trees()
        a = single.search_batch(batch, temperatures=temps)
        b = dual.search_batch(batch, temperatures=temps)
        assert torch.equal(a.chosen_action_indices, b.chosen_action_indices)
        assert torch.equal(a.policy_dense, b.policy_dense) and torch.equal(a.root_value, b.root_value)
        assert torch.equal(a.terminal_mask, b.terminal_mask) and torch.equal(a.legal_mask, b.legal_mask)
        # second move with subtree reuse on both
        from liuzhou_amd import v0_core
        nxt = batch._map(lambda t: t.clone())
        plies = torch.zeros(B, dtype=torch.int64, device=DEV); done = torch.zeros(B, dtype=torch.bool, device=DEV)
        v0_core.self_play_step_inplace(*nxt.tensors(), plies, done, torch.arange(B, device=DEV), a.chosen_action_codes,
                                       a.terminal_mask, a.chosen_valid_mask, 512, 2.0)
        a2 = single.search_batch(nxt, temperatures=temps, active=~done)
        b2 = dual.search_batch(nxt, temperatures=temps, active=~done)
        assert torch.equal(a2.chosen_action_indices, b2.chosen_action_indices) and torch.equal(a2.policy_dense, b2.policy_dense)
    assert dual.leaf_evals == single.leaf_evals


def test_tree_engine_degenerate_sizes():
    """One game / one simulation, two games on two streams, zero games through the C ABI: nothing faults, visits add up."""
    _need_gpu()
    import ctypes as C
    from liuzhou_amd import _lib as L
    from liuzhou_amd.mcts_gpu import GpuStateBatch
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import DualStreamTreeMCTS, LzTreeDesc, PortableTreeMCTS, TreeEngine
    torch.manual_seed(1)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    for cls, B, sims in ((PortableTreeMCTS, 1, 1), (PortableTreeMCTS, 3, 2), (DualStreamTreeMCTS, 2, 1), (DualStreamTreeMCTS, 5, 3)):
        m = cls(net, B, sims, DEV, sample_moves=False, add_dirichlet_noise=False, reuse_tree=True)
        st = GpuStateBatch.initial(DEV, B)
        plies = torch.zeros(B, dtype=torch.int64, device=DEV); done = torch.zeros(B, dtype=torch.bool, device=DEV)
        from liuzhou_amd import v0_core
        for mv in range(3):
            out = m.search_batch(st, temperatures=torch.ones(B, device=DEV))
            assert bool(out.chosen_valid_mask.all()) and torch.allclose(out.policy_dense.sum(1), torch.ones(B, device=DEV), atol=1e-5)
            v0_core.self_play_step_inplace(*st.tensors(), plies, done, torch.arange(B, device=DEV), out.chosen_action_codes,
                                           out.terminal_mask, out.chosen_valid_mask, 512, 2.0)
    # zero games: every entry point is a no-op that reports success
    eng = TreeEngine(1, 4, DEV)
    d = LzTreeDesc()
    C.memmove(C.byref(d), C.byref(eng.desc), C.sizeof(LzTreeDesc))
    d.num_games = 0
    z = torch.zeros(8, device=DEV)
    assert L.lib().lz_tree_begin(C.byref(d), None) == 0 and L.lib().lz_tree_select(C.byref(d), None) == 0
    assert L.lib().lz_tree_advance(C.byref(d), None, None, L.i64(4), None, None, None) == 0
    assert L.lib().lz_tree_expand(C.byref(d), C.c_int(1), None, None, None, L.ptr(z), L.ptr(z), None, L.i64(0), C.c_float(0.25), None) == 0
    # invalid descriptors are refused, not launched
    d.num_games = 1
    d.nodes = None
    assert L.lib().lz_tree_begin(C.byref(d), None) == -1
    d.nodes = eng.desc.nodes
    d.node_cap = 600000
    assert L.lib().lz_tree_advance(C.byref(d), None, None, L.i64(4), None, None, None) == -2      # > 524288 nodes: unsupported
    d.node_cap = eng.desc.node_cap
    d.edge_chunk = 100                                           # not a power of two
    assert L.lib().lz_tree_begin(C.byref(d), None) == -1
    d.edge_chunk = eng.desc.edge_chunk
    eng2 = TreeEngine(1, 40, DEV)                                # 42 nodes x 72 children > one chunk of 1024 edges
    C.memmove(C.byref(d), C.byref(eng2.desc), C.sizeof(LzTreeDesc))
    d.chunk_cap = 1                                              # a chunk list that cannot hold the node arena's worst case
    assert L.lib().lz_tree_advance(C.byref(d), None, None, L.i64(4), None, None, None) == -1


def test_auto_sizing_node_arena_and_edge_pool():
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine, auto_pool_chunks, auto_reuse_factor, chunk_cap_for, REUSE_FACTOR_CAP
    assert auto_reuse_factor(256, 200, DEV) == REUSE_FACTOR_CAP   # plenty of memory: the cap
    assert auto_reuse_factor(256, 4000, DEV) <= (65536 - 4002) / 4000 + 1e-9     # 65 536 nodes per game at most
    free, _ = torch.cuda.mem_get_info(torch.device(DEV))
    games = int(free // (3 * 1024 * 1024))                        # ~3 MB of free memory per game, a tenth of it for nodes
    assert 1.0 <= auto_reuse_factor(games, 800, DEV) < REUSE_FACTOR_CAP
    eng = TreeEngine(64, 50, DEV, reuse_factor=-1.0)
    assert eng.reuse_factor == REUSE_FACTOR_CAP and eng.node_cap == 50 + 2 + int(REUSE_FACTOR_CAP) * 50
    # <= 64 games: the pool holds the worst case (72 children everywhere), an allocation cannot fail
    assert eng.pool_chunks == 64 * chunk_cap_for(eng.node_cap, eng.edge_chunk)
    assert eng.chunk_cap * (eng.edge_chunk - 71) >= eng.node_cap * 72
    # C3's shape: the mean case with head-room, within the 2^31 pool indices (64 GB) -- not 16 384 worst cases
    n = auto_pool_chunks(16384, 800, 32802, 1024, DEV)
    assert n * 1024 * 32 <= 64 * 2**30 and n < 16384 * chunk_cap_for(32802, 1024) // 4
    st = eng.pool_status()
    assert st["free"] == st["chunks"] == st["fewest_free"] and st["refused_expansions"] == 0


def test_edge_pool_exhaustion_is_counted_not_a_fault():
    """A pool that is too small: the expansions that find no chunk are refused and counted, the search goes on (the leaf
    stays unexpanded, its value is backed up), every root still gets `sims` visits and nothing is written out of bounds."""
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    from tests.tree_parity import hash_evaluator, unpack_packed
    from oracle import lz_oracle as O
    B, sims = 8, 60
    eng = TreeEngine(B, sims, DEV, 1.0, edge_chunk=128, pool_chunks=B * 3)      # ~10 expansions per game, 61 wanted
    guard = eng.buf["edges"].clone()
    eng.set_roots(to_gpu_batch(O.initial_states(B), DEV))
    eng.begin()
    for s in range(sims + 1):
        if s:
            eng.select()
        leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
        pri, val = hash_evaluator(leaf)
        eng.expand(is_root=(s == 0), values=torch.from_numpy(val).to(DEV), priors220=torch.from_numpy(pri).to(DEV))
    eng.finish(torch.ones(B, device=DEV), None, sample_moves=False)
    st = eng.pool_status()
    assert st["refused_expansions"] > 0 and st["free"] == 0 and st["fewest_free"] == 0
    assert (eng.buf["root_visits"].cpu().numpy() == sims).all()
    assert (eng.child_visits.sum(1).cpu().numpy() == sims).all() and bool(eng.chosen_valid.all())
    assert int(eng.buf["n_chunks"].sum()) == eng.pool_chunks and guard.shape == eng.buf["edges"].shape
    eng.begin()                                                   # a fresh search gives every chunk back
    assert eng.pool_status()["free"] == eng.pool_chunks


def test_a_chunk_is_reserved_for_every_fresh_root():
    """The smallest pool the ABI accepts (one chunk per game): every root still expands -- a chunk stays reserved for each
    fresh root until its first expansion (pool_stats[2]), so deeper expansions of other games cannot drain the pool under
    a root, which would otherwise end its search with an all-zero policy -- and a pool smaller than the batch is refused."""
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    from liuzhou_amd import _lib as L
    from tests.tree_parity import hash_evaluator, unpack_packed
    from oracle import lz_oracle as O
    import ctypes as C
    B, sims = 16, 20
    eng = TreeEngine(B, sims, DEV, 1.0, edge_chunk=128, pool_chunks=B)
    eng.set_roots(to_gpu_batch(O.initial_states(B), DEV))
    eng.begin()
    assert int(eng.buf["pool_stats"][2]) == B                    # B live roots are waiting for their chunk
    for s in range(sims + 1):
        if s:
            eng.select()
        leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
        pri, val = hash_evaluator(leaf)
        eng.expand(is_root=(s == 0), values=torch.from_numpy(val).to(DEV), priors220=torch.from_numpy(pri).to(DEV))
        if s == 0:
            assert int(eng.buf["pool_stats"][2]) == 0 and int(eng.buf["pool_top"]) == 0
    eng.finish(torch.ones(B, device=DEV), None, sample_moves=False)
    assert (eng.child_count.cpu().numpy() == 36).all() and bool(eng.chosen_valid.all())
    assert torch.allclose(eng.policy_dense.sum(1), torch.ones(B, device=DEV), atol=1e-5)
    assert eng.pool_status()["refused_expansions"] > 0
    small = TreeEngine(B, sims, DEV, 1.0, edge_chunk=128, pool_chunks=B - 1)
    with torch.cuda.device(DEV):
        assert L.lib().lz_tree_begin(C.byref(small.desc), L.stream_ptr(torch.device(DEV))) == -1      # LZ_ERR_ARG


def test_root_chunk_reservation_follows_the_tree_state():
    """pool_stats[2] is the number of live fresh roots that have not expanded: a second begin of pending
    roots, a begin over expanded trees, inactive games and a root without a legal move leave no chunk reserved for
    nobody; reset_run keeps it."""
    _need_gpu()
    from liuzhou_amd.tree_engine import TreeEngine
    from tests.tree_parity import hash_evaluator, unpack_packed
    B, sims = 16, 8
    eng = TreeEngine(B, sims, DEV, 1.0, edge_chunk=128, pool_chunks=4 * B)
    res = lambda: int(eng.buf["pool_stats"][2])

    def expand_root():
        leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
        pri, val = hash_evaluator(leaf)
        eng.expand(is_root=True, values=torch.from_numpy(val).to(DEV), priors220=torch.from_numpy(pri).to(DEV))

    eng.set_roots(to_gpu_batch(O.initial_states(B), DEV))
    eng.begin(); eng.begin()
    assert res() == B                                              # counted once, not twice
    expand_root()
    assert res() == 0
    eng.begin()                                                    # over expanded trees: fresh roots again
    assert res() == B
    eng.buf["active"][: B // 2] = 0
    eng.begin()                                                    # half of the pending roots become inactive
    assert res() == B // 2
    eng.buf["active"].fill_(1)
    eng.begin(); eng.begin()
    assert res() == B
    expand_root()
    assert res() == 0 and int(eng.buf["pool_top"]) == 4 * B - B
    # roots that are not finished but have no legal move (g2) take no chunk and hold no reservation afterwards
    z = load("g2_edges.npz")
    st = states(z, "s")
    mask, _ = O.encode_actions(st)
    stuck = np.flatnonzero(mask.sum(axis=1) == 0)
    if stuck.size:
        pick = np.resize(stuck, B)
        sub = {f: np.ascontiguousarray(np.asarray(st[f])[pick]) for f in FIELDS}
        eng.set_roots(to_gpu_batch(sub, DEV))
        eng.begin()
        pending = res()
        expand_root()
        assert res() == 0, (pending, res())


@pytest.mark.parametrize("batch_k,sims", [(16, 50), (4, 30), (16, 200)])
def test_gpu_wave_batched_search_matches_oracle(batch_k, sims):
    """The legacy search's waves (src/mcts.py `batch_K` leaves per tree and wave, no virtual loss; oracle pinned by
    g13): same leaves in the same order in every wave, bit-exact visit counts / priors / picks over three moves with
    subtree reuse."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests.tree_parity import run_injected_wave_parity
    games = 24 if sims == 200 else 48
    eng, waves, short = run_injected_wave_parity("cuda:0", num_games=games, sims=sims, batch_k=batch_k, moves=3,
                                                 seed=batch_k + sims)
    assert waves >= 3 * -(-sims // batch_k)


def test_gpu_wave_search_follows_descents_deeper_than_48_levels():
    """A pseudo-network that puts nearly all prior mass on the first legal action and values every position 0 makes the
    tree one long line that grows by one level per wave: with 320 simulations in waves of 4 from the empty board the walks go
    about 80 levels deep (until round 5 the wave kernel stopped following a descent at level 48; its level stack now covers
    the 144 plies a game can last).  Same
    leaves in the same order as the oracle in every wave, bit-exact visit counts."""
    _need_gpu()
    from tests.tree_parity import run_injected_wave_parity, hash_evaluator

    def line_evaluator(st):
        pri, _ = hash_evaluator(st)
        mask, _ = O.encode_actions(st)
        first = np.argmax(mask, axis=1)
        pri = (pri * np.float32(1e-3)).astype(np.float32)
        pri[np.arange(pri.shape[0]), first] = np.float32(1.0)
        return pri, np.zeros(pri.shape[0], np.float32)

    eng, waves, short = run_injected_wave_parity("cuda:0", sims=320, batch_k=4, moves=1, seed=2, with_noise=False,
                                                 states=O.initial_states(4), evaluator=line_evaluator)
    depth = int(eng.wbuf["path_len"].max().item())
    assert depth > 48, depth                                   # the last wave's deepest leaf path


@pytest.mark.parametrize("limit", [3, 9])
def test_gpu_wave_walks_give_up_after_the_references_backtrack_limit(limit):
    """src/mcts.py:337,371-414: a walk gives up after MAX_BACKTRACK_STEPS (128) upward moves, and since every walk of a wave
    restarts at the root and replays the earlier ones, the wave ends there.  With the limit lowered to 3 / 9 on narrow
    trees (roots with at most three legal actions, 32 leaves per wave) it bites in most waves: the kernel, which continues
    one traversal and counts its upward moves cumulatively, still collects leaf for leaf what the oracle's literal
    restart-and-ban walks collect -- fewer leaves per wave than an unlimited traversal finds."""
    _need_gpu()
    from tests.tree_parity import run_injected_wave_parity
    z = load("g1_rules.npz")
    st = states(z, "s")
    mask, _ = O.encode_actions(st)
    narrow = np.flatnonzero((mask.sum(axis=1) >= 1) & (mask.sum(axis=1) <= 3))
    pick = narrow[np.random.default_rng(limit).permutation(narrow.size)[:40]]
    sub = {f: np.ascontiguousarray(np.asarray(st[f])[pick]) for f in FIELDS}
    try:
        eng, waves, short = run_injected_wave_parity("cuda:0", sims=120, batch_k=32, moves=2, seed=limit, states=sub,
                                                     max_backtrack=limit)
        _, waves_free, short_free = run_injected_wave_parity("cuda:0", sims=120, batch_k=32, moves=2, seed=limit, states=sub)
    finally:
        O.lib().lzo_set_max_backtrack(128)
    assert short > short_free and waves > waves_free, (short, short_free, waves, waves_free)


def test_gpu_wave_batched_search_on_narrow_trees_needs_extra_rounds():
    """Roots with at most three legal actions: a wave often finds fewer open leaves than batch_k, so the budget is used
    up over more rounds than ceil(sims / batch_k) -- still leaf for leaf the oracle's waves."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests.tree_parity import run_injected_wave_parity
    z = load("g1_rules.npz")
    st = states(z, "s")
    mask, _ = O.encode_actions(st)
    narrow = np.flatnonzero((mask.sum(axis=1) >= 1) & (mask.sum(axis=1) <= 3))
    assert narrow.size >= 16
    pick = narrow[np.random.default_rng(2).permutation(narrow.size)[:32]]
    sub = {f: np.ascontiguousarray(np.asarray(st[f])[pick]) for f in FIELDS}
    eng, waves, short = run_injected_wave_parity("cuda:0", sims=40, batch_k=16, moves=2, seed=9, states=sub)
    assert short > 0 and waves > 2 * 3


def test_gpu_wave_search_with_fused_network_compact_batch_equals_slot_major_protocol():
    """lz_tree_search_waves (C++ loop; the leaves that need the network go through a compact, device-counted batch)
    gives the same trees as the split-phase protocol in which the fused network evaluates all batch_k * B slots in
    place: the compact list, its row indirection and the network's independence of the batch position."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import TreeEngine
    dev, K, sims, B = "cuda:0", 16, 60, 96
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    z = load("g1_rules.npz")
    st = states(z, "s")
    idx = np.random.default_rng(11).integers(0, st["board"].shape[0], B)
    sub = {f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}
    engines = [TreeEngine(B, sims, dev, 1.0, batch_k=K) for _ in range(2)]
    for e in engines:
        e.set_roots(to_gpu_batch(sub, dev))
    a, b = engines
    a.search(net, sims)
    a.finish_waves(net, sims)
    b.begin()
    lp1, lp2, lpm, _, val = net.forward_packed(b.buf["leaf_state"])
    b.expand(is_root=True, values=val, heads=(lp1, lp2, lpm))
    first = True
    for _ in range(4096):
        b.select_wave(sims, reset_budget=first)
        first = False
        lp1, lp2, lpm, _, val = net.forward_packed(b.wbuf["leaf_state"])
        b.expand_wave(values=val, heads=(lp1, lp2, lpm), slot_major=True)
        if int(b.wbuf["unfinished"].item()) == 0:
            break
    temps = torch.ones((B,), dtype=torch.float32, device=dev)
    for e in engines:
        e.finish(temps, None)
    va, pa = engine_visits(a)
    vb, pb = engine_visits(b)
    assert np.array_equal(va, vb) and np.array_equal(pa, pb)
    live = ~a.terminal_mask.cpu().numpy()
    assert live.any() and (va[live].sum(axis=1) == sims).all()
    assert torch.equal(a.chosen_index, b.chosen_index) and torch.equal(a.root_value, b.root_value)


# ---- the production launch path (what bench.py and the product runners execute) vs the oracle ------------------------
@pytest.mark.parametrize("dual", [False, True])
def test_production_search_path_replayed_in_oracle(dual):
    """lz_tree_search + lz_tree_search_continue exactly as bench.py drives them: fused expand + select kernel with the
    priors formed in the kernel from the three head rows, fused b6c64 network in the loop, one hipGraph per move,
    subtree reuse, Philox root noise, sampled moves; one engine and two engines on two streams.  128 games x 200 sims
    x 3 consecutive moves: same leaf at every simulation, bit-identical visit counts / W sums / priors at the root,
    in-kernel softmax within 1e-6 of the oracle's projection of the same head rows."""
    _need_gpu()
    from tests.tree_parity import run_production_parity
    mcts, tot = run_production_parity(DEV, "b6c64", num_games=128, sims=200, moves=3, dual=dual, seed=21)
    assert mcts.use_graph and not mcts.graph_retry_off
    assert tot["kept"] > 0, "no game kept a subtree: the continued search was not exercised"
    assert tot["evals"] > 128 * 200 * 2
    print(f"production parity dual={dual}: {tot}")


def test_persistent_search_kernel_replayed_in_oracle_and_equal_to_the_launch_pairs(monkeypatch):
    """lz_tree_search_persistent (csrc/lz_search.hip, opt-in: one launch per move, a workgroup owns 8 games): the same
    production-parity replay in the oracle -- graph replay, subtree reuse, Philox noise, sampled moves, a game count that is
    not a multiple of the 8 games per workgroup -- and bit-identical root edge records to the per-simulation launch pairs
    (the same device functions run the network pass and the tree steps in both)."""
    _need_gpu()
    from tests.tree_parity import run_production_parity, root_edges, EDGE_LOGICAL
    ref, _ = run_production_parity(DEV, "b6c64", num_games=77, sims=96, moves=3, seed=31)
    assert not ref.engine.persistent
    monkeypatch.setenv("LZ_TREE_PERSISTENT", "1")
    got, tot = run_production_parity(DEV, "b6c64", num_games=77, sims=96, moves=3, seed=31)
    assert got.engine.persistent and got.engine.persistent_ok(got.net) and got.use_graph and not got.graph_retry_off
    assert tot["kept"] > 0
    for x, y in zip(root_edges(got.engine), root_edges(ref.engine)):
        assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL) and ((x["child"] >= 0) == (y["child"] >= 0)).all()
    assert torch.equal(got.engine.chosen_index, ref.engine.chosen_index)
    # the 128-channel net and the fp32 parity mode are refused by the entry point and stay on the launch pairs
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    torch.manual_seed(1)
    big = FusedNet(ChessNet(**MODEL_CONFIGS["b10c128"]).eval().to(DEV))
    assert not got.engine.persistent_ok(big)
    assert not got.engine.persistent_ok(FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV), precision="fp32"))


def test_production_search_path_c3_arithmetic():
    """C3's arithmetic and arena sizes on a small population: 10x128 network, 800 simulations per move, 64 games, two
    consecutive moves (the second one continues kept subtrees of several hundred nodes)."""
    _need_gpu()
    from tests.tree_parity import run_production_parity
    mcts, tot = run_production_parity(DEV, "b10c128", num_games=64, sims=800, moves=2, dual=False, seed=22,
                                      reuse_factor=4.0)
    assert mcts.use_graph and tot["kept"] > 0
    print(f"production parity C3 arithmetic: {tot}")


@pytest.mark.parametrize("model_name,games,sims", [("b6c64", 77, 96), ("b10c128", 64, 800)])
def test_compact_evaluation_lists_equal_the_dense_search(model_name, games, sims):
    """`compact_evals` (LzTreeDesc.live_*): per simulation only the leaves that need the network are evaluated, through a
    device-side list and lz_net_forward_packed_counted_f16.  The production replay in the oracle passes unchanged (same
    leaf at every simulation, bit-identical visit counts / W sums / priors, network rows == a direct launch on the same
    states), the root edge records equal the dense search's byte for byte, and every launched evaluation is a consumed one."""
    _need_gpu()
    from tests.tree_parity import run_production_parity, root_edges, EDGE_LOGICAL
    dense, _ = run_production_parity(DEV, model_name, num_games=games, sims=sims, moves=3, seed=31)
    assert not dense.engine.compact_evals
    got, tot = run_production_parity(DEV, model_name, num_games=games, sims=sims, moves=3, seed=31, compact_evals=True)
    assert got.engine.compact_evals and got.use_graph and not got.graph_retry_off and tot["kept"] > 0
    for x, y in zip(root_edges(got.engine), root_edges(dense.engine)):
        assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL) and ((x["child"] >= 0) == (y["child"] >= 0)).all()
    assert torch.equal(got.engine.chosen_index, dense.engine.chosen_index)
    assert torch.equal(got.engine.policy_dense, dense.engine.policy_dense)
    assert got.consumed_evals == dense.consumed_evals                      # the same leaves needed the network ...
    assert got.leaf_evals == got.consumed_evals                            # ... and nothing else was launched (both counts
    #                                                                        include the two-simulation warm-up before capture)
    print(f"compact lists {model_name}: launched {got.leaf_evals} of the dense search's {dense.leaf_evals}")


def test_alternating_launch_forms_over_consecutive_moves_equal_the_dense_search():
    """An engine built for both launch forms (what the runner gets when a full launch is several network passes per CU)
    switches between dense launches and compact lists from move to move -- fresh search, continued searches with kept
    subtrees, each form captured as its own graph on first use -- and ends with the dense engine's trees, byte for byte."""
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import PortableTreeMCTS
    from tests.tree_parity import root_edges, EDGE_LOGICAL, to_gpu_batch
    from tests.golden_utils import load, states as gstates
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    st_all = gstates(load("g1_rules.npz"), "s")
    idx = np.random.default_rng(5).integers(0, st_all["board"].shape[0], 96)
    cur = {f: np.ascontiguousarray(np.asarray(st_all[f])[idx]) for f in FIELDS}
    kw = dict(exploration_weight=1.0, add_dirichlet_noise=True, sample_moves=True, reuse_tree=True, reuse_factor=4.0, seed=99)
    dense = PortableTreeMCTS(net, 96, 64, DEV, compact_evals=False, **kw)
    both = PortableTreeMCTS(net, 96, 64, DEV, compact_evals=True, **kw)
    both.compact_default = False                                   # as the automatic choice builds it: dense unless told otherwise
    temps = torch.ones((96,), device=DEV)
    for move, form in enumerate([False, True, True, False, True]):
        batch = to_gpu_batch(cur, DEV)
        a = dense.search_batch(batch, temperatures=temps)
        b = both.search_batch(batch, temperatures=temps, compact=form)
        torch.cuda.synchronize()
        for x, y in zip(root_edges(both.engine), root_edges(dense.engine)):
            assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL), (move, form)
        assert torch.equal(a.chosen_action_indices, b.chosen_action_indices) and torch.equal(a.policy_dense, b.policy_dense)
        # play the picked moves (the oracle's transition), so that the next search continues kept subtrees
        from oracle import lz_oracle as O
        pick = a.chosen_action_indices.cpu().numpy()
        nxt = [O.apply_index(O.state_from_batch(cur, i), int(pick[i])) if pick[i] >= 0 else O.state_from_batch(cur, i) for i in range(96)]
        cur = O.batch_from_states(nxt)
    assert both.list_searches == 3 and both.use_graph and not both.graph_retry_off
    assert len(both._graphs) >= 3                                  # fresh-dense, continued-lists, continued-dense


def test_production_search_direct_launches_equal_graph_replay():
    """The same search with direct launches (LZ_TREE_GRAPH=off path) and as a replayed hipGraph: identical trees."""
    _need_gpu()
    from tests.tree_parity import run_production_parity, root_edges, EDGE_LOGICAL
    a, _ = run_production_parity(DEV, "b6c64", num_games=48, sims=64, moves=2, seed=23, use_graph=True)
    b, _ = run_production_parity(DEV, "b6c64", num_games=48, sims=64, moves=2, seed=23, use_graph=False)
    for x, y in zip(root_edges(a.engine), root_edges(b.engine)):
        assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL) and ((x["child"] >= 0) == (y["child"] >= 0)).all()


@pytest.mark.parametrize("model,compact", [("b6c64", False), ("b10c128", True)])
def test_split_step_on_two_waves_builds_the_same_trees_as_the_one_wave_step(monkeypatch, model, compact):
    """Round 6: launches of at most 8 192 games run the simulation step on TWO waves per game (the first expands the previous
    leaf while the second backs up and descends; csrc/lz_engine.hip::tree_expand_select_split_kernel).  Both forms replay
    against the oracle (run_production_parity compares every simulation's leaf) and end with byte-identical root records;
    also with the compact evaluation lists, over consecutive moves with kept subtrees."""
    _need_gpu()
    from tests.tree_parity import run_production_parity, root_edges, EDGE_LOGICAL
    kw = dict(num_games=96, sims=72, moves=3, seed=31, compact_evals=compact)
    monkeypatch.setenv("LZ_TREE_SPLIT", "1")
    a, _ = run_production_parity(DEV, model, **kw)
    monkeypatch.setenv("LZ_TREE_SPLIT", "0")
    b, _ = run_production_parity(DEV, model, **kw)
    for x, y in zip(root_edges(a.engine), root_edges(b.engine)):
        assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL) and ((x["child"] >= 0) == (y["child"] >= 0)).all()
    assert torch.equal(a.engine.buf["n_nodes"], b.engine.buf["n_nodes"]) and torch.equal(a.engine.buf["root_w"], b.engine.buf["root_w"])


@pytest.mark.parametrize("flat_priors", [False, True])
def test_single_precision_argmax_of_the_descent_never_changes_a_tree(monkeypatch, flat_priors):
    """Round 6 (opt-in, LZ_TREE_F32SEL=1): a level's argmax is taken in fp32 when its best candidate leads by more than 100 x
    the fp32 error bound, in double otherwise (near ties, exact ties, NaN).  Double everywhere (the default) and the fp32 form build the same trees
    byte for byte over 3 moves with kept subtrees; `flat_priors`: a network whose policy heads are zero -- every prior of a
    node equal, so unvisited children tie EXACTLY and the double path has to break the ties by index."""
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import PortableTreeMCTS
    from tests.tree_parity import root_edges, EDGE_LOGICAL, to_gpu_batch
    from liuzhou_amd import v0_core
    torch.manual_seed(20260314)
    m = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    if flat_priors:
        with torch.no_grad():
            for conv in (m.policy_head.out_pos1, m.policy_head.out_pos2, m.policy_head.out_mark):
                conv.weight.zero_()
    net = FusedNet(m.to(DEV))
    st_all = states(load("g1_rules.npz"), "s")
    idx = np.random.default_rng(11).integers(0, st_all["board"].shape[0], 192)
    engines = []
    for flag in ("1", "0"):                                        # (the fp32 argmax is opt-in: measured slower on random-init nets)
        monkeypatch.setenv("LZ_TREE_F32SEL", flag)
        batch = to_gpu_batch({f: np.ascontiguousarray(np.asarray(st_all[f])[idx]) for f in FIELDS}, DEV)
        e = PortableTreeMCTS(net, 192, 160, DEV, add_dirichlet_noise=True, sample_moves=True, reuse_tree=True, seed=5)
        temps = torch.ones(192, device=DEV)
        picks = []
        for mv in range(3):
            out = e.search_batch(batch, temperatures=temps)
            picks.append(out.chosen_action_indices.clone())
            plies = torch.zeros(192, dtype=torch.int64, device=DEV); done = torch.zeros(192, dtype=torch.bool, device=DEV)
            v0_core.self_play_step_inplace(*batch.tensors(), plies, done, torch.arange(192, device=DEV),
                                           out.chosen_action_codes.clone(), out.terminal_mask.clone(),
                                           out.chosen_valid_mask.clone(), 512, 2.0)
        engines.append((e, picks, out.policy_dense.clone()))
    (a, pa, pola), (b, pb, polb) = engines
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    assert torch.equal(pola, polb)
    for x, y in zip(root_edges(a.engine), root_edges(b.engine)):
        assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL)
    assert torch.equal(a.engine.buf["n_nodes"], b.engine.buf["n_nodes"]) and torch.equal(a.engine.buf["root_w"], b.engine.buf["root_w"])


def test_graph_capture_failure_falls_back_to_direct_launches(monkeypatch):
    """A failed capture (the reference retries a failed finalize-graph capture with the graph off,
    v1/python/self_play_worker.py:434-442) must not lose the move: same result as the direct path, flag recorded."""
    _need_gpu()
    from tests.tree_parity import run_production_parity, root_edges, EDGE_LOGICAL
    ref, _ = run_production_parity(DEV, "b6c64", num_games=32, sims=32, moves=2, seed=24, use_graph=False)
    monkeypatch.setenv("LZ_TREE_GRAPH_FAULT", "capture")
    got, _ = run_production_parity(DEV, "b6c64", num_games=32, sims=32, moves=2, seed=24, use_graph=True)
    assert got.graph_retry_off and not got.use_graph
    for x, y in zip(root_edges(got.engine), root_edges(ref.engine)):
        assert all(x[f].tobytes() == y[f].tobytes() for f in EDGE_LOGICAL) and ((x["child"] >= 0) == (y["child"] >= 0)).all()


def test_device_rng_equals_oracle_and_is_independent_of_the_batch_split():
    """lz_rng_gamma / lz_rng_uniform (Philox4x32-10 keyed by seed, counter = game id / ply / purpose / index) against
    the numpy oracle; and the search built on them plays the SAME moves with noise and sampling on whether the games run
    as one batch on one stream or as two halves on two streams."""
    _need_gpu()
    from oracle import rng_oracle as R
    from liuzhou_amd.game_rng import GameRng, PURPOSE_PICK
    rng = GameRng(300, DEV, seed=(5 << 32) | 99, game_offset=1000)
    rng.ply.copy_(torch.arange(300, device=DEV) % 144)
    out = torch.zeros((300, 80), device=DEV); u = torch.zeros((300,), device=DEV)
    rng.gamma_into(out, 0.3, 72); rng.uniform_into(u, PURPOSE_PICK)
    game, ply = rng.game.cpu().numpy(), rng.ply.cpu().numpy()
    assert np.array_equal(u.cpu().numpy(), R.uniform(rng.seed, game, ply, R.PURPOSE_PICK))
    np.testing.assert_allclose(out[:, :72].cpu().numpy(), R.gamma(rng.seed, game, ply, 72, 0.3), rtol=1e-4, atol=1e-30)
    assert not out[:, 72:].any()

    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import DualStreamTreeMCTS, PortableTreeMCTS
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    z = load("g1_rules.npz")
    st = states(z, "s")
    idx = np.random.default_rng(9).integers(0, st["board"].shape[0], 96)
    batch = to_gpu_batch({f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}, DEV)
    temps = torch.ones(96, device=DEV)
    kw = dict(add_dirichlet_noise=True, sample_moves=True, seed=4242, reuse_tree=True, reuse_factor=2.0)
    one = PortableTreeMCTS(net.variant(half_workgroups=True), 96, 40, DEV, **kw)
    two = DualStreamTreeMCTS(net, 96, 40, DEV, **kw)
    from liuzhou_amd import v0_core
    s1 = batch._map(lambda t: t.clone()); s2 = batch._map(lambda t: t.clone())
    for mv in range(3):
        o1 = one.search_batch(s1, temperatures=temps)
        o2 = two.search_batch(s2, temperatures=temps)
        assert torch.equal(o1.chosen_action_indices, o2.chosen_action_indices), mv
        assert torch.equal(o1.policy_dense, o2.policy_dense), mv
        for s, o in ((s1, o1), (s2, o2)):
            plies = torch.zeros(96, dtype=torch.int64, device=DEV); done = torch.zeros(96, dtype=torch.bool, device=DEV)
            v0_core.self_play_step_inplace(*s.tensors(), plies, done, torch.arange(96, device=DEV),
                                           o.chosen_action_codes.clone(), o.terminal_mask.clone(),
                                           o.chosen_valid_mask.clone(), 512, 2.0)
    # another seed plays other games
    a = PortableTreeMCTS(net, 96, 40, DEV, **kw).search_batch(batch, temperatures=temps).chosen_action_indices.clone()
    b = PortableTreeMCTS(net, 96, 40, DEV, **kw).search_batch(batch, temperatures=temps).chosen_action_indices.clone()
    c = PortableTreeMCTS(net, 96, 40, DEV, **{**kw, "seed": 4243}).search_batch(batch, temperatures=temps).chosen_action_indices
    assert torch.equal(a, b) and not torch.equal(a, c)      # same seed: same games; another seed: other games


def test_production_search_path_on_edge_states():
    """The reference tests' representative / terminal states (g2: finished games, a MARK_SELECTION state without a legal
    move, forced-removal states ...) through the production launch path, two moves: terminal roots stay inactive, no-legal
    leaves back up -1, everything else is replayed bit for bit."""
    _need_gpu()
    from tests.tree_parity import run_production_parity
    z = load("g2_edges.npz")
    st = states(z, "s")
    for dual in (False, True):
        mcts, tot = run_production_parity(DEV, "b6c64", sims=40, moves=2, dual=dual, seed=31,
                                          states={f: np.ascontiguousarray(np.asarray(st[f])) for f in FIELDS})
        assert tot["evals"] > 0
