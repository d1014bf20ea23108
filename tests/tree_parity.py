"""Shared helpers for the tree-engine parity tests (GPU engine vs oracle/lz_oracle.c)."""
import numpy as np
import torch

from oracle import lz_oracle as O
from tests.golden_utils import FIELDS

FULL = (1 << 36) - 1


def unpack_packed(packed: np.ndarray):
    """int64[B,4] packed records (liuzhou_amd/csrc/lz_rules.h:pack) -> state dict of numpy arrays."""
    p = packed.astype(np.uint64)
    B = p.shape[0]
    w0, w1, w2, w3 = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    bits = lambda w: ((w[:, None] >> np.arange(36, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64)
    black, white = bits(w0), bits(w1)
    st = O.empty_states(B)
    st["board"] = (black - white).astype(np.int8).reshape(B, 6, 6)
    st["marks_black"] = bits(w2).astype(bool).reshape(B, 6, 6)
    st["marks_white"] = bits(w3).astype(bool).reshape(B, 6, 6)
    f = lambda sh, m: ((w0 >> np.uint64(sh)) & np.uint64(m)).astype(np.int64)
    st["move_count"] = f(36, 0xFF); st["moves_since_capture"] = f(44, 0x3F); st["phase"] = f(50, 7)
    st["current_player"] = np.where(f(53, 1) == 1, -1, 1).astype(np.int64)
    st["forced_removals_done"] = f(54, 3); st["pending_marks_required"] = f(56, 3)
    st["pending_marks_remaining"] = f(58, 3); st["pending_captures_required"] = f(60, 3)
    st["pending_captures_remaining"] = f(62, 3)
    return st


def hash_evaluator(states):
    """Deterministic pseudo-network: priors220 (strictly positive on every index) + value in (-1,1),
    a pure function of the state bytes -> bit-identical on both sides."""
    B = states["board"].shape[0]
    key = np.zeros(B, np.uint64)
    flat = np.concatenate([np.asarray(states["board"]).reshape(B, 36).astype(np.int64) + 1,
                           np.asarray(states["marks_black"]).reshape(B, 36).astype(np.int64),
                           np.asarray(states["marks_white"]).reshape(B, 36).astype(np.int64),
                           np.stack([np.asarray(states[f]).astype(np.int64) for f in FIELDS[3:]], axis=1) + 2], axis=1)
    for j in range(flat.shape[1]):
        key = (key * np.uint64(6364136223846793005) + flat[:, j].astype(np.uint64) + np.uint64(1442695040888963407))
    a = np.arange(220, dtype=np.uint64)[None, :]
    h = (key[:, None] ^ (a * np.uint64(0x9E3779B97F4A7C15)))
    h = (h ^ (h >> np.uint64(31))) * np.uint64(0xBF58476D1CE4E5B9)
    h = h ^ (h >> np.uint64(29))
    pri = ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24) + 0.01).astype(np.float32)
    pri = (pri / pri.sum(1, keepdims=True, dtype=np.float32)).astype(np.float32)
    hv = (key ^ (key >> np.uint64(33))) * np.uint64(0xFF51AFD7ED558CCD)
    val = (((hv >> np.uint64(40)).astype(np.float64) / float(1 << 24)) * 1.8 - 0.9).astype(np.float32)
    return pri, val


def to_gpu_batch(st, device):
    from liuzhou_amd.mcts_gpu import GpuStateBatch
    ts = []
    for f in FIELDS:
        a = np.ascontiguousarray(st[f])
        dt = np.int8 if f == "board" else (bool if f.startswith("marks") else np.int64)
        ts.append(torch.from_numpy(a.astype(dt)).to(device))
    return GpuStateBatch(*ts)


def engine_visits(engine):
    cnt = engine.child_count.cpu().numpy()
    act = engine.child_action.cpu().numpy(); vis = engine.child_visits.cpu().numpy(); pri = engine.child_prior.cpu().numpy()
    out_v = np.zeros((engine.B, 220), np.int32); out_p = np.zeros((engine.B, 220), np.float32)
    for g in range(engine.B):
        k = int(cnt[g])
        out_v[g, act[g, :k]] = vis[g, :k]; out_p[g, act[g, :k]] = pri[g, :k]
    return out_v, out_p


def run_injected_parity(device, num_games=64, sims=64, seed=0, states=None, c=1.0, noise=None, eps=0.25,
                        temperature=1.0):
    """GPU tree engine vs oracle tree, both driven by `hash_evaluator`; returns the engine for inspection.
    Asserts bit-exact visit counts, priors, picks, and close policy / root value."""
    from liuzhou_amd.tree_engine import TreeEngine
    from tests.golden_utils import load, states as gstates
    if states is None:
        z = load("g1_rules.npz")
        st_all = gstates(z, "s")
        rng = np.random.default_rng(seed)
        idx = rng.integers(0, st_all["board"].shape[0], num_games)
        states = {f: np.ascontiguousarray(np.asarray(st_all[f])[idx]) for f in FIELDS}
    B = states["board"].shape[0]
    eng = TreeEngine(B, sims, device, c)
    eng.set_roots(to_gpu_batch(states, device))
    eng.begin()
    trees = [O.OracleTree(O.state_from_batch(states, i), c) for i in range(B)]
    nz_dev = None if noise is None else torch.from_numpy(noise.astype(np.float32)).to(device)

    def complete(is_root):
        kind = eng.buf["leaf_kind"].cpu().numpy()
        leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
        pend = [t.prepare_root() if is_root else t.select() for t in trees]
        want_kind = np.array([1 if p else 0 for p in pend])
        got_kind = (kind == 1).astype(int)
        assert np.array_equal(got_kind, want_kind), "GPU and oracle disagree on which games need an evaluation"
        need = np.nonzero(want_kind)[0]
        if need.size:
            o_states = O.batch_from_states([trees[i].pending_state() for i in need])
            for f in FIELDS:
                a = np.asarray(leaf[f])[need].reshape(need.size, -1).astype(np.int64)
                b = np.asarray(o_states[f]).reshape(need.size, -1).astype(np.int64)
                assert np.array_equal(a, b), f"leaf state field {f} differs"
        pri, val = hash_evaluator(leaf)
        for k, i in enumerate(need):
            nz = None
            if is_root and noise is not None:
                nz = noise[i]
            trees[i].complete(pri[i], float(val[i]), nz, eps)
        eng.expand(is_root=is_root, values=torch.from_numpy(val).to(device),
                   priors220=torch.from_numpy(pri).to(device), noise=nz_dev if is_root else None, epsilon=eps)

    complete(True)
    for _ in range(sims):
        eng.select()
        complete(False)
    temps = torch.full((B,), float(temperature), dtype=torch.float32, device=device)
    eng.finish(temps, None)
    got_v, got_p = engine_visits(eng)
    chosen = eng.chosen_index.cpu().numpy()
    pol = eng.policy_dense.cpu().numpy()
    rv = eng.root_value.cpu().numpy()
    term = eng.terminal_mask.cpu().numpy()
    for i, t in enumerate(trees):
        if t.root_terminal():
            assert term[i], i
            continue
        idx, vis, vs, pr, pl = t.root_children()
        want = np.zeros(220, np.int32); want[idx] = vis
        assert np.array_equal(got_v[i], want), (i, np.abs(got_v[i] - want).sum())
        wp = np.zeros(220, np.float32); wp[idx] = pr
        assert np.array_equal(got_p[i], wp), i
        assert int(vis.sum()) == sims
        p = np.zeros(220, np.float32); p[idx] = O.policy_from_visits(vis, temperature)
        np.testing.assert_allclose(pol[i], p, atol=1e-6, rtol=0)
        want_rv = t.root_value_sum() / max(1, t.root_visits())
        assert abs(float(rv[i]) - want_rv) < 1e-6
        # deterministic pick: most visits -> Q -> prior -> lowest index
        rp = t.root_player()
        q = np.where(vis > 0, np.where(pl == rp, vs, -vs) / np.maximum(vis, 1), 0.0).astype(np.float32)
        cand = np.nonzero(vis == vis.max())[0]
        cand = cand[np.abs(q[cand] - q[cand].max()) <= 1e-6]
        cand = cand[np.abs(pr[cand] - pr[cand].max()) <= 1e-8]
        assert int(chosen[i]) == int(idx[cand.min()]), i
    return eng


def run_injected_reuse_parity(device, num_games=48, sims=48, moves=4, seed=3, c=1.0, with_noise=True, eps=0.25,
                              reuse_factor=4.0, edge_chunk=None):
    """Tree reuse (a21): `moves` consecutive searches per game; after each one the deterministic pick is played on
    both sides, the oracle promotes the child with `advance` (portable_mcts.py:74-87) and the GPU engine with
    lz_tree_advance.  Bit-exact visit counts / priors after every move, including the re-noised kept roots."""
    from liuzhou_amd.tree_engine import TreeEngine, OUT_CAP
    from tests.golden_utils import load, states as gstates
    z = load("g1_rules.npz")
    st_all = gstates(z, "s")
    rng = np.random.default_rng(seed)
    idx0 = rng.integers(0, st_all["board"].shape[0], num_games)
    states = {f: np.ascontiguousarray(np.asarray(st_all[f])[idx0]) for f in FIELDS}
    B = num_games
    # `edge_chunk`: small chunks of the edge pool (128 records) make every few expansions take a new chunk and every
    # compaction of a kept subtree re-pack its runs across many chunk boundaries
    eng = TreeEngine(B, sims, device, c, reuse_factor=reuse_factor, **({} if edge_chunk is None else {"edge_chunk": edge_chunk}))
    cur = [O.state_from_batch(states, i) for i in range(B)]
    trees = [O.OracleTree(cur[i], c) for i in range(B)]
    kept_total = 0
    for mv in range(moves):
        batch = O.batch_from_states(cur)
        eng.set_roots(to_gpu_batch(batch, device))
        if mv == 0:
            eng.begin()
        else:
            eng.advance()                       # played action = the last finish()'s pick
        noise = rng.gamma(0.3, 1.0, size=(B, OUT_CAP)).astype(np.float32) + np.float32(1e-6) if with_noise else None
        nz_dev = None if noise is None else torch.from_numpy(noise).to(device)

        kind = eng.buf["leaf_kind"].cpu().numpy()
        leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
        pend = [t.prepare_root() for t in trees]
        assert np.array_equal(kind == 1, np.array(pend)), f"move {mv}: fresh / kept roots differ"
        if mv > 0:
            kept = [(not p) and t.root_children()[0].size > 0 for p, t in zip(pend, trees)]
            assert np.array_equal(kind == 3, np.array(kept)), f"move {mv}: kept roots differ"
            kept_total += int(np.sum(kept))
        pri, val = hash_evaluator(leaf)
        for i, t in enumerate(trees):
            if pend[i]:
                t.complete(pri[i], float(val[i]), None if noise is None else noise[i], eps)
            elif noise is not None and t.root_children()[0].size > 0:
                t.root_noise(noise[i], eps)
        eng.expand(is_root=True, values=torch.from_numpy(val).to(device), priors220=torch.from_numpy(pri).to(device),
                   noise=nz_dev, epsilon=eps)
        for _ in range(sims):
            eng.select()
            kind = eng.buf["leaf_kind"].cpu().numpy()
            leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
            pend = [t.select() for t in trees]
            assert np.array_equal(kind == 1, np.array(pend)), f"move {mv}: pending evaluations differ"
            pri, val = hash_evaluator(leaf)
            for i, t in enumerate(trees):
                if pend[i]:
                    t.complete(pri[i], float(val[i]))
            eng.expand(is_root=False, values=torch.from_numpy(val).to(device), priors220=torch.from_numpy(pri).to(device))
        temps = torch.full((B,), 0.1, dtype=torch.float32, device=device)
        eng.finish(temps, None)
        got_v, got_p = engine_visits(eng)
        chosen = eng.chosen_index.cpu().numpy()
        rv = eng.root_value.cpu().numpy()
        term = eng.terminal_mask.cpu().numpy()
        for i, t in enumerate(trees):
            if t.root_terminal():
                assert term[i] and chosen[i] == -1, (mv, i)
                continue
            idx, vis, vs, pr, pl = t.root_children()
            want = np.zeros(220, np.int32); want[idx] = vis
            assert np.array_equal(got_v[i], want), (mv, i, np.abs(got_v[i] - want).sum())
            wp = np.zeros(220, np.float32); wp[idx] = pr
            assert np.array_equal(got_p[i], wp), (mv, i)
            assert abs(float(rv[i]) - t.root_value_sum() / max(1, t.root_visits())) < 1e-6
            from oracle.selfplay_oracle import deterministic_pick
            pick = deterministic_pick(idx, vis, vs, pr, pl, t.root_player())
            assert int(chosen[i]) == pick, (mv, i)
            cur[i] = O.apply_index(cur[i], pick)
            if not t.advance(pick):
                trees[i] = O.OracleTree(cur[i], c)
    assert eng.reuse_dropped.tolist() == [0, 0]
    return eng, kept_total


def run_injected_wave_parity(device, num_games=48, sims=50, batch_k=16, moves=3, seed=5, c=1.0, with_noise=True, eps=0.25,
                             reuse_factor=4.0, states=None, max_backtrack=None, evaluator=None):
    """The legacy search's waves (src/mcts.py batch_K, oracle: lzo_tree_select_wave / complete_wave, pinned by g13) on
    the GPU engine: `moves` consecutive searches with subtree reuse, both sides driven by `hash_evaluator`.  Every wave
    must collect the same leaves in the same order (the leaf states are compared), and after every search the root
    visit counts, priors, value and pick must agree bit for bit."""
    from liuzhou_amd.tree_engine import TreeEngine, OUT_CAP
    from oracle.selfplay_oracle import deterministic_pick
    from tests.golden_utils import load, states as gstates
    rng = np.random.default_rng(seed)
    evaluate = evaluator or hash_evaluator
    if states is None:
        z = load("g1_rules.npz")
        st_all = gstates(z, "s")
        idx0 = rng.integers(0, st_all["board"].shape[0], num_games)
        states = {f: np.ascontiguousarray(np.asarray(st_all[f])[idx0]) for f in FIELDS}
    B, K = states["board"].shape[0], int(batch_k)
    eng = TreeEngine(B, sims, device, c, reuse_factor=reuse_factor, batch_k=K, max_backtrack_steps=max_backtrack or 0)
    O.lib().lzo_set_max_backtrack(int(max_backtrack or 128))      # (the caller restores 128: tests/test_gpu_tree.py)
    cur = [O.state_from_batch(states, i) for i in range(B)]
    trees = [O.OracleTree(cur[i], c) for i in range(B)]
    waves_total = short_waves = 0
    for mv in range(moves):
        eng.set_roots(to_gpu_batch(O.batch_from_states(cur), device))
        if mv == 0:
            eng.begin()
        else:
            eng.advance()
        noise = rng.gamma(0.3, 1.0, size=(B, OUT_CAP)).astype(np.float32) + np.float32(1e-6) if with_noise else None
        nz_dev = None if noise is None else torch.from_numpy(noise).to(device)
        kind = eng.buf["leaf_kind"].cpu().numpy()
        leaf = unpack_packed(eng.buf["leaf_state"].cpu().numpy())
        pend = [t.prepare_root() for t in trees]
        assert np.array_equal(kind == 1, np.array(pend)), f"move {mv}: fresh / kept roots differ"
        pri, val = evaluate(leaf)
        for i, t in enumerate(trees):
            if pend[i]:
                t.complete(pri[i], float(val[i]), None if noise is None else noise[i], eps)
            elif noise is not None and t.root_children()[0].size > 0:
                t.root_noise(noise[i], eps)
        eng.expand(is_root=True, values=torch.from_numpy(val).to(device), priors220=torch.from_numpy(pri).to(device),
                   noise=nz_dev, epsilon=eps)
        done = [0] * B
        first = True
        while True:
            eng.select_wave(sims, reset_budget=first)
            first = False
            kinds = eng.wbuf["leaf_kind"].cpu().numpy().reshape(K, B)
            slots = unpack_packed(eng.wbuf["leaf_state"].cpu().numpy())            # [K*B] slot-major
            pri, val = evaluate(slots)
            progressed = False
            for i, t in enumerate(trees):
                if done[i] >= sims or t.root_terminal():
                    assert not kinds[:, i].any(), (mv, i)
                    continue
                got = t.select_wave(min(K, sims - done[i]))
                found = int((kinds[:, i] != 0).sum())
                assert found == got and kinds[:found, i].all(), (mv, i, found, got)
                short_waves += int(got < min(K, sims - done[i]))
                done[i] = sims if got == 0 else done[i] + got
                progressed = progressed or got > 0
                # the oracle's evaluation list = the GPU's "expand" slots that have a legal move, in slot order
                ws = t.wave_states()
                rows = []
                for j in range(found):
                    if kinds[j, i] == 1:
                        cs = O.state_from_batch({f: np.asarray(slots[f])[j * B + i: j * B + i + 1] for f in FIELDS}, 0)
                        if O.legal_indices_py(cs):
                            rows.append(j * B + i)
                assert len(rows) == len(ws), (mv, i, len(rows), len(ws))
                if ws:
                    want = O.batch_from_states(ws)
                    for f in FIELDS:
                        a = np.asarray(slots[f])[rows].reshape(len(rows), -1).astype(np.int64)
                        b = np.asarray(want[f]).reshape(len(rows), -1).astype(np.int64)
                        assert np.array_equal(a, b), f"move {mv} game {i}: wave leaf field {f} differs"
                    t.complete_wave(pri[rows], val[rows])
            waves_total += 1
            assert (int(eng.wbuf["unfinished"].item()) > 0) == any(
                done[i] < sims and not trees[i].root_terminal() for i in range(B)), mv
            eng.expand_wave(values=torch.from_numpy(val).to(device), priors220=torch.from_numpy(pri).to(device))
            if not progressed or all(done[i] >= sims or trees[i].root_terminal() for i in range(B)):
                break
        temps = torch.full((B,), 0.1, dtype=torch.float32, device=device)
        eng.finish(temps, None)
        got_v, got_p = engine_visits(eng)
        chosen = eng.chosen_index.cpu().numpy()
        rv = eng.root_value.cpu().numpy()
        term = eng.terminal_mask.cpu().numpy()
        for i, t in enumerate(trees):
            if t.root_terminal():
                assert term[i] and chosen[i] == -1, (mv, i)
                continue
            idx, vis, vs, pr, pl = t.root_children()
            want = np.zeros(220, np.int32); want[idx] = vis
            assert np.array_equal(got_v[i], want), (mv, i, np.abs(got_v[i] - want).sum())
            wp = np.zeros(220, np.float32); wp[idx] = pr
            assert np.array_equal(got_p[i], wp), (mv, i)
            assert abs(float(rv[i]) - t.root_value_sum() / max(1, t.root_visits())) < 1e-6
            pick = deterministic_pick(idx, vis, vs, pr, pl, t.root_player())
            assert int(chosen[i]) == pick, (mv, i)
            cur[i] = O.apply_index(cur[i], pick)
            if not t.advance(pick):
                trees[i] = O.OracleTree(cur[i], c)
    assert eng.reuse_dropped.tolist() == [0, 0]
    return eng, waves_total, short_waves


# ---------------------------------------------------------------------------------------------------------------------
# Production launch path: lz_tree_search / lz_tree_search_continue (fused expand + select kernel, priors formed in the
# kernel from the three head rows, fused network kernel in the loop, hipGraph replay, one or two streams, subtree reuse,
# Philox root noise and pick uniforms) replayed in the oracle from the trace the expand kernel leaves (LzTreeDesc.trace_*).
# ---------------------------------------------------------------------------------------------------------------------
EDGE_DT = np.dtype([("W", "<f8"), ("P", "<f4"), ("n_info", "<u4"), ("child", "<i4"), ("cbegin", "<i4"), ("act", "u1"),
                    ("cn", "u1"), ("pad", "V6")])
NODE_DT = np.dtype([("state", "<i8", (4,)), ("edge_begin", "<i4"), ("nedges", "<i4"), ("parent", "<i4"), ("old_begin", "<i4")])
# what a root edge record says about the search (everything but the place of the child's run in the engine's edge pool,
# which depends on the order in which the games' waves took their chunks)
EDGE_LOGICAL = ("W", "P", "n_info", "act", "cn")


def game_tree(engine, g):
    """(nodes, runs): game g's node records and, per node, its edge run read from the engine's pool."""
    nn = int(engine.buf["n_nodes"][g])
    nodes = engine.buf["nodes"].view(engine.B, engine.node_cap, 6)[g, :nn].contiguous().cpu().numpy().view(NODE_DT).reshape(nn)
    pool = engine.buf["edges"]
    runs = []
    for nd in nodes:
        e0, n = int(nd["edge_begin"]), int(nd["nedges"])
        runs.append(pool[e0:e0 + n].contiguous().cpu().numpy().view(EDGE_DT).reshape(n) if n > 0 else np.zeros(0, EDGE_DT))
    return nodes, runs


def root_edges(engine):
    """Root edge records of every game, read straight from the pool: list of structured arrays (EDGE_DT)."""
    B = engine.B
    nodes = engine.buf["nodes"].view(B, engine.node_cap, 6)[:, 0].contiguous().cpu().numpy().view(NODE_DT).reshape(B)
    pool = engine.buf["edges"]
    out = []
    for g in range(B):
        ne, e0 = int(nodes["nedges"][g]), int(nodes["edge_begin"][g])
        if ne <= 0:
            out.append(np.zeros(0, EDGE_DT))
            continue
        out.append(pool[e0:e0 + ne].contiguous().cpu().numpy().view(EDGE_DT).reshape(ne))
    return out


def _states_equal(leaf, rows, want_states, what):
    want = O.batch_from_states(want_states)
    for f in FIELDS:
        a = np.asarray(leaf[f])[rows].reshape(len(rows), -1).astype(np.int64)
        b = np.asarray(want[f]).reshape(len(rows), -1).astype(np.int64)
        assert np.array_equal(a, b), f"{what}: leaf state field {f} differs from the oracle's pending state"


def replay_part_in_oracle(part, trees, move, c_eps, check_net=None, float_tol=1e-6):
    """One searched move of one engine (`part`: PortableTreeMCTS built with trace=True) against the oracle trees of its
    games.  Float part: the priors the kernel formed from the head rows are within `float_tol` of the oracle's
    projection of the same rows.  Integer part: fed the kernel's own per-step priors and values, the oracle must ask for
    the same leaf state at every step and end with bit-identical visit counts, value sums (f64) and priors at the root."""
    e = part.engine
    S, n = part.sims, e.B
    tr = {k: v.cpu().numpy() for k, v in e.trace.items()}
    kind, val = tr["trace_kind"], tr["trace_value"]
    leaf = unpack_packed(tr["trace_leaf"].reshape(-1, 4))
    heads, pri = tr["trace_heads"], tr["trace_priors"]
    noise = part._noise_buf.cpu().numpy()
    stats = dict(evals=0, kept=0, max_prior_err=0.0)
    for s in range(S + 1):
        pend = [t.prepare_root() if s == 0 else t.select() for t in trees]
        assert np.array_equal(kind[s] == 1, np.array(pend)), \
            f"move {move} step {s}: GPU and oracle disagree on which games need an evaluation"
        if s == 0:
            kept = np.array([(not p) and (not t.root_terminal()) and t.root_children()[0].size > 0
                             for p, t in zip(pend, trees)])
            assert np.array_equal(kind[0] == 3, kept), f"move {move}: kept roots differ"
            stats["kept"] += int(kept.sum())
        need = np.nonzero(pend)[0]
        if need.size:
            ps = [trees[i].pending_state() for i in need]
            _states_equal(leaf, s * n + need, ps, f"move {move} step {s}")
            mask = np.zeros((need.size, 220), bool)
            for k, cs in enumerate(ps):
                mask[k, O.legal_indices_py(cs)] = True
            has = mask.any(1)
            want, _ = O.project_policy(heads[s, need, 0:36], heads[s, need, 36:72], heads[s, need, 72:108], mask)
            err = np.abs(pri[s, need][has] - want[has]).max() if has.any() else 0.0
            stats["max_prior_err"] = max(stats["max_prior_err"], float(err))
            assert err <= float_tol, f"move {move} step {s}: in-kernel head -> prior softmax off by {err}"
            assert not pri[s, need][~mask].any(), "prior mass on an illegal action"
            if check_net is not None and has.any():       # (a leaf without a legal move is backed up as -1: its rows are not read)
                check_net(s, need[has], tr["trace_leaf"][s, need[has]], heads[s, need[has]], val[s, need[has]])
            stats["evals"] += int(need.size)
        for i in need:
            trees[i].complete(pri[s, i], float(val[s, i]), noise[i] if (s == 0 and c_eps is not None) else None,
                              0.25 if c_eps is None else c_eps)
        if s == 0 and c_eps is not None:
            for i in np.nonzero(kept)[0]:
                trees[i].root_noise(noise[i], c_eps)
    # ---- after the search: the root's statistics, bit for bit ----
    edges = root_edges(e)
    term = e.terminal_mask.cpu().numpy()
    rv = e.root_value.cpu().numpy()
    for i, t in enumerate(trees):
        if t.root_terminal():
            assert term[i], (move, i)
            continue
        idx, vis, vs, pr, _pl = t.root_children()
        E = edges[i]
        assert np.array_equal(E["act"].astype(np.int64), idx.astype(np.int64)), (move, i)
        assert np.array_equal((E["n_info"] & 0xFFFFFF).astype(np.int64), vis.astype(np.int64)), \
            (move, i, "visit counts differ")
        assert np.array_equal(E["W"].view(np.uint64), vs.astype(np.float64).view(np.uint64)), (move, i, "W sums differ")
        assert np.array_equal(E["P"].view(np.uint32), pr.astype(np.float32).view(np.uint32)), (move, i, "priors differ")
        assert abs(float(rv[i]) - t.root_value_sum() / max(1, t.root_visits())) < 1e-6
    return stats


def run_production_parity(device, model_name="b6c64", num_games=128, sims=200, moves=3, dual=False, seed=0, use_graph=True,
                          noise=True, temperature=1.0, reuse_factor=4.0, rng_seed=777, states=None, compact_evals=False):
    """bench.py's search (PortableTreeMCTS / DualStreamTreeMCTS with the fused network, hipGraph, subtree reuse, Philox
    noise and sampled moves) over `moves` consecutive moves from a mixed-phase batch, replayed in the oracle."""
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import DualStreamTreeMCTS, PortableTreeMCTS
    from tests.golden_utils import load, states as gstates
    dev = torch.device(device)
    torch.manual_seed(20260314)
    module = ChessNet(**MODEL_CONFIGS[model_name]).eval().to(dev)
    net = FusedNet(module)
    net32 = FusedNet(module, precision="fp32")       # an independent kernel (fp32 operands, <= 1e-5 of the reference's module)
    if states is None:
        z = load("g1_rules.npz")
        st_all = gstates(z, "s")
        rng = np.random.default_rng(seed)
        idx0 = rng.integers(0, st_all["board"].shape[0], num_games)
        states = {f: np.ascontiguousarray(np.asarray(st_all[f])[idx0]) for f in FIELDS}
    B = num_games = int(np.asarray(states["board"]).shape[0])
    kw = dict(exploration_weight=1.0, add_dirichlet_noise=noise, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
              sample_moves=True, use_graph=use_graph, reuse_tree=True, reuse_factor=reuse_factor, trace=True, seed=rng_seed,
              compact_evals=bool(compact_evals))
    mcts = (DualStreamTreeMCTS if dual else PortableTreeMCTS)(net, B, sims, dev, **kw)
    parts = list(zip(mcts.bounds, mcts.parts)) if dual else [((0, B), mcts)]
    cur = [O.state_from_batch(states, i) for i in range(B)]
    trees = [O.OracleTree(cur[i], 1.0) for i in range(B)]
    totals = dict(evals=0, kept=0, max_prior_err=0.0, net_max_err=0.0)

    def check_net_factory(part):
        def check(s, need, packed, heads, values):
            # the network launch inside the captured search evaluated exactly these states: an independent launch of the
            # same kernel configuration on them gives the same head rows and values
            lp1, lp2, lpm, _, v = part.net.forward_packed(torch.from_numpy(np.ascontiguousarray(packed)).to(dev))
            got = torch.cat([lp1, lp2, lpm], dim=1).cpu().numpy()
            d = max(float(np.abs(got - heads).max()), float(np.abs(v.cpu().numpy() - values).max()))
            totals["net_max_err"] = max(totals["net_max_err"], d)
            assert d <= 1e-6, f"step {s}: network rows inside the captured search differ from a direct launch by {d}"
            # ... and not only with itself: the fp32-operand kernel (lz_net_f32.hip, pinned to the reference's own outputs
            # by g9) evaluates the same states -- fp16 rounding only (observed 9.4e-5 on log-probs, 2e-6 on values)
            q1, q2, qm, _, qv = net32.forward_packed(torch.from_numpy(np.ascontiguousarray(packed)).to(dev))
            ref = torch.cat([q1, q2, qm], dim=1).cpu().numpy()
            live = ref > -12.0
            d32 = max(float(np.abs(heads - ref)[live].max()), float(np.abs(qv.cpu().numpy() - values).max()) * 10.0)
            totals["net_fp32_err"] = max(totals.get("net_fp32_err", 0.0), d32)
            assert d32 <= 1e-3, f"step {s}: network rows inside the captured search are {d32} away from the fp32 kernel"
        return check

    for mv in range(moves):
        batch = to_gpu_batch(O.batch_from_states(cur), dev)
        temps = torch.full((B,), float(temperature), dtype=torch.float32, device=dev)
        out = mcts.search_batch(batch, temperatures=temps)
        torch.cuda.synchronize(dev)
        chosen = out.chosen_action_indices.cpu().numpy()
        pol = out.policy_dense.cpu().numpy()
        for (a, b), part in parts:
            st = replay_part_in_oracle(part, trees[a:b], mv, 0.25 if noise else None,
                                       check_net=check_net_factory(part) if mv == 0 else None)
            for k in ("evals", "kept"):
                totals[k] += st[k]
            totals["max_prior_err"] = max(totals["max_prior_err"], st["max_prior_err"])
            assert part.engine.reuse_dropped.tolist() == [0, 0]
            u = part._uniforms.cpu().numpy()
            for i in range(a, b):
                t = trees[i]
                if t.root_terminal():
                    assert chosen[i] == -1
                    continue
                idx, vis, _vs, _pr, _pl = t.root_children()
                want = np.zeros(220, np.float32); want[idx] = O.policy_from_visits(vis, temperature)
                np.testing.assert_allclose(pol[i], want, atol=1e-6, rtol=0)
                # sampled pick: inverse CDF of the selection policy over the children in ascending action order
                k = int(np.nonzero(idx == chosen[i])[0][0])
                cum = np.cumsum(want[idx].astype(np.float64))
                assert want[idx][k] > 0 and cum[k] > u[i - a] - 1e-6 and (k == 0 or cum[k - 1] <= u[i - a] + 1e-6), \
                    (mv, i, "pick is not the inverse-CDF sample of the policy")
        for i in range(B):
            if trees[i].root_terminal():
                # no move was played: the engine finds no child to keep and starts this game's next search from a fresh
                # root (a runner would have ended the game here) -- so does the replay
                trees[i] = O.OracleTree(cur[i], 1.0)
                continue
            pick = int(chosen[i])
            cur[i] = O.apply_index(cur[i], pick)
            if not trees[i].advance(pick):
                trees[i] = O.OracleTree(cur[i], 1.0)
    return mcts, totals
