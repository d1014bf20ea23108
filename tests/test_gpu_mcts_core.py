"""GPU: the `MCTSConfig` / `MCTSCore` / `InferenceEngine` / `EvalBatcher` / `TorchScriptRunner` adapters (v0_core class
surface)."""
import types

import numpy as np
import pytest
import torch

from tests.golden_utils import load, states, FIELDS
from tests.tree_parity import to_gpu_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _state_like(st, i):
    """A Python object shaped like src/game_state.py's GameState (what v0/python/mcts.py hands to set_root_state)."""
    mk = lambda a: [(r, c) for r in range(6) for c in range(6) if a[r][c]]
    return types.SimpleNamespace(
        board=np.asarray(st["board"][i]).reshape(6, 6).astype(int).tolist(),
        marked_black=mk(np.asarray(st["marks_black"][i]).reshape(6, 6)),
        marked_white=mk(np.asarray(st["marks_white"][i]).reshape(6, 6)),
        phase=types.SimpleNamespace(value=int(st["phase"][i])), current_player=int(st["current_player"][i]),
        pending_marks_required=int(st["pending_marks_required"][i]), pending_marks_remaining=int(st["pending_marks_remaining"][i]),
        pending_captures_required=int(st["pending_captures_required"][i]),
        pending_captures_remaining=int(st["pending_captures_remaining"][i]),
        forced_removals_done=int(st["forced_removals_done"][i]), move_count=int(st["move_count"][i]))


def test_mcts_core_adapter_equals_the_batch_engine_and_keeps_subtrees():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.tree_engine import PortableTreeMCTS
    torch.manual_seed(20260314)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV)
    engine = v0_core.InferenceEngine(model, device=DEV, dtype="float16", batch_size=64)
    assert engine.graph_enabled and engine.batch_size == 64 and engine.device == DEV
    z = load("g1_rules.npz")
    st = states(z, "s")
    rng = np.random.default_rng(3)
    for i in rng.integers(0, st["board"].shape[0], 6).tolist():
        one = {f: np.ascontiguousarray(np.asarray(st[f])[i:i + 1]) for f in FIELDS}
        one["moves_since_capture"][:] = 0                      # the reference's coercion does not carry this field
        cfg = v0_core.MCTSConfig()
        cfg.num_simulations, cfg.device, cfg.exploration_weight = 48, DEV, 1.0
        core = v0_core.MCTSCore(cfg)
        core.set_inference_engine(engine)
        core.set_root_state(_state_like(st, i))
        core.run_simulations(48)
        ref = PortableTreeMCTS(engine.fused, 1, 48, DEV, add_dirichlet_noise=False, sample_moves=False, use_graph=False)
        out = ref.search_batch(to_gpu_batch(one, DEV), temperatures=torch.ones(1, device=DEV))
        stats = core.get_root_children_stats()
        k = int(ref.engine.child_count.item())
        assert [s["action_index"] for s in stats] == ref.engine.child_action[0, :k].tolist()
        assert [int(s["visit_count"]) for s in stats] == ref.engine.child_visits[0, :k].tolist()
        assert core.root_visit_count == 48 and abs(core.root_value - float(out.root_value.item())) < 1e-6
        pol = dict(core.get_policy(1.0))
        dense = out.policy_dense[0].cpu().numpy()
        for a, p in pol.items():
            assert abs(p - float(dense[a])) < 1e-6
        assert abs(sum(pol.values()) - 1.0) < 1e-9
        hot = core.get_policy(0.0)
        assert sorted(p for _, p in hot)[-1] == 1.0 and sum(p for _, p in hot) == 1.0
        # tree reuse: the played child's statistics survive advance_root (mcts_core.cpp:815-829)
        best = max(stats, key=lambda s: s["visit_count"])
        core.advance_root(best["action_index"])
        kept = core.root_visit_count
        assert kept == best["visit_count"]
        core.run_simulations(16)
        assert core.root_visit_count == kept + 16
        assert core.get_eval_stats()["eval_leaves"] >= 48 + 16
        core.advance_root(219 if best["action_index"] != 219 else 218)      # not a child: reset
        assert core.get_policy(1.0) == []


def test_mcts_core_with_a_python_forward_callback():
    """set_forward_callback (module.cpp:1177-1196): any callable returning (log_p1, log_p2, log_pmc, value)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to(DEV)
    calls = []

    def forward(x):
        calls.append(int(x.shape[0]))
        with torch.inference_mode():
            lp1, lp2, lpm, raw = model(x.to(DEV))
        return lp1, lp2, lpm, bucket_logits_to_scalar(raw)

    cfg = v0_core.MCTSConfig()
    cfg.device, cfg.num_simulations, cfg.add_dirichlet_noise = DEV, 32, True
    core = v0_core.MCTSCore(cfg)
    core.set_forward_callback(forward)
    st = states(load("g1_rules.npz"), "s")
    core.set_root_state(_state_like(st, 10))
    core.run_simulations(32)
    assert len(calls) >= 2 and core.root_visit_count == 32
    pol = core.get_policy(1.0)
    assert pol and abs(sum(p for _, p in pol) - 1.0) < 1e-9
    with pytest.raises(RuntimeError, match="null"):
        core.set_torchscript_runner(None)


def test_mcts_core_against_the_oracle_tree():
    """f4 parity proper: `MCTSCore` (one game on the device tree engine, split-phase protocol, a Python forward callback)
    against `oracle.OracleTree` driven by the SAME evaluations -- the oracle must ask for the same position at every
    simulation and end with the same children: action indices and visit counts bit-exact, value sums equal as doubles
    (same additions in the same order), priors within 1e-6 (the kernel forms them from the head rows itself), over a
    search, an advance_root that keeps the subtree, and a second search.  Semantics: variant P (v1/python/portable_mcts.py),
    which `MCTSCore` documents as its deliberate deviation from v0/src/mcts/mcts_core.cpp:270-278."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
    from oracle import lz_oracle as O
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to(DEV)
    log = []

    def forward(x):
        with torch.inference_mode():
            lp1, lp2, lpm, raw = model(x.to(DEV))
            val = bucket_logits_to_scalar(raw)
        log.append((x.detach().cpu().numpy().copy(), lp1.cpu().numpy().copy(), lp2.cpu().numpy().copy(),
                    lpm.cpu().numpy().copy(), float(val.reshape(-1)[0])))
        return lp1, lp2, lpm, val

    st = states(load("g1_rules.npz"), "s")
    for i in (3, 400, 1200, 2000):
        one = {f: np.ascontiguousarray(np.asarray(st[f])[i:i + 1]) for f in FIELDS}
        one["moves_since_capture"][:] = 0                      # the reference's coercion does not carry this field
        cfg = v0_core.MCTSConfig()
        cfg.device, cfg.num_simulations, cfg.exploration_weight, cfg.add_dirichlet_noise = DEV, 40, 1.0, False
        core = v0_core.MCTSCore(cfg)
        core.set_forward_callback(forward)
        del log[:]
        core.set_root_state(_state_like(st, i))
        core.run_simulations(40)
        tree = O.OracleTree(O.state_from_batch(one, 0), 1.0)

        def replay(calls):
            """Feed the oracle the evaluations MCTSCore consumed, in order; returns the number of positions compared."""
            used, k = 0, 0
            steps = [True] + [False] * len(calls)              # the root step, then one selection per remaining call
            for is_root in steps:
                if k >= len(calls) and not is_root:
                    break
                pend = tree.prepare_root() if is_root else tree.select()
                if is_root and not pend:
                    continue                                   # kept (or terminal) root: MCTSCore did not evaluate it either
                planes, lp1, lp2, lpm, val = calls[k]
                k += 1
                if not pend:
                    continue                                   # terminal leaf: MCTSCore evaluated it, nobody reads the result
                cs = tree.pending_state()
                want_planes = O.states_to_model_input(O.batch_from_states([cs]))
                assert np.array_equal(planes.reshape(1, 11, 6, 6), want_planes), f"game {i} call {k}: another leaf"
                mask = np.zeros((1, 220), bool)
                mask[0, O.legal_indices_py(cs)] = True
                pri, _ = O.project_policy(lp1.reshape(1, 36), lp2.reshape(1, 36), lpm.reshape(1, 36), mask)
                tree.complete(pri[0], val)
                used += 1
            assert k == len(calls), "MCTSCore made more evaluations than the oracle's search has steps"
            return used

        assert replay(list(log)) >= 30

        def compare():
            stats = core.get_root_children_stats()
            idx, vis, vs, pr, pl = tree.root_children()
            assert [s["action_index"] for s in stats] == idx.tolist()
            assert [int(s["visit_count"]) for s in stats] == vis.tolist()
            np.testing.assert_allclose([s["prior"] for s in stats], pr, atol=1e-6, rtol=0)
            # value sums: MCTSCore reports them from the root mover's side, the oracle from the child mover's
            rp = int(core._root.current_player.item())
            want = [float(w) if int(p) == rp else -float(w) for w, p in zip(vs, pl)]
            assert [s["value_sum"] for s in stats] == want
            return stats

        stats = compare()
        assert core.root_visit_count == 40
        # advance_root keeps the played child's subtree on both sides; the next search continues it
        best = max(stats, key=lambda s: (s["visit_count"], -s["action_index"]))
        core.advance_root(best["action_index"])
        tree.advance(int(best["action_index"]))
        del log[:]
        core.run_simulations(24)
        assert replay(list(log)) >= (0 if tree.root_terminal() else 10)
        compare()


def test_mcts_core_root_noise_is_fresh_for_every_root():
    """the reference's generator is stateful (mcts_core.cpp:132,316) -- every set_root_state / advance_root
    draws fresh Dirichlet noise.  Two consecutive set_root_state calls on the same position must not mix the same noise."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to(DEV)

    def forward(x):
        with torch.inference_mode():
            lp1, lp2, lpm, raw = model(x.to(DEV))
        return lp1, lp2, lpm, bucket_logits_to_scalar(raw)

    st = states(load("g1_rules.npz"), "s")
    cfg = v0_core.MCTSConfig()
    cfg.device, cfg.num_simulations, cfg.add_dirichlet_noise, cfg.seed = DEV, 8, True, 4242
    core = v0_core.MCTSCore(cfg)
    core.set_forward_callback(forward)
    priors = []
    for _ in range(3):
        core.set_root_state(_state_like(st, 10))
        core.run_simulations(1)
        priors.append([s["prior"] for s in core.get_root_children_stats()])
    assert len(priors[0]) > 1
    assert priors[0] != priors[1] and priors[1] != priors[2] and priors[0] != priors[2]
    # a simulation budget beyond the arena limit (524 288 nodes since round 6; 65 536 before) is refused with a clear message
    # instead of an arena error deep inside the engine
    big = v0_core.MCTSConfig()
    big.device, big.num_simulations = DEV, 200000
    core2 = v0_core.MCTSCore(big)
    core2.set_forward_callback(forward)
    with pytest.raises(ValueError, match="524288"):
        core2.set_root_state(_state_like(st, 10))


def test_mcts_core_searches_17000_simulations_and_keeps_the_subtree_in_a_quarter_million_node_arena():
    """Round 6: arenas beyond 65 536 nodes per game (the reference's tree is unbounded, v1/cpp/portable_mcts.cpp:739-769; until
    round 5 `MCTSCore` refused num_simulations > 16 383).  17 000 simulations of one game need an arena of 272 002 nodes;
    `advance_root` compacts it with the one-wave-per-workgroup form of lz_tree_advance (96 KB of LDS marks) and the next search
    continues the kept subtree: visit counts add up exactly."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    torch.manual_seed(3)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    cfg = v0_core.MCTSConfig()
    cfg.device, cfg.num_simulations, cfg.add_dirichlet_noise, cfg.seed = DEV, 17000, False, 1
    core = v0_core.MCTSCore(cfg)
    core.set_inference_engine(v0_core.InferenceEngine(model, DEV, "float16", 64))
    st = states(load("g1_rules.npz"), "s")
    core.set_root_state(_state_like(st, 10))
    assert core._engine.node_cap == 4 * 68000 + 2 > 65536
    core.run_simulations(17000)
    stats = core.get_root_children_stats()
    assert core.root_visit_count == 17000 and sum(s["visit_count"] for s in stats) == 17000
    best = max(stats, key=lambda s: (s["visit_count"], -s["action_index"]))
    assert best["visit_count"] > 1000
    core.advance_root(best["action_index"])
    ts = core.get_tree_stats()
    assert ts["reuse_dropped"] == 0 and ts["reuse_pruned"] == 0
    core.run_simulations(50)
    kept = best["visit_count"] - 1                          # the child's visits minus its own expansion = visits of ITS children
    assert sum(s["visit_count"] for s in core.get_root_children_stats()) == kept + 50


def test_torchscript_runner_and_eval_batcher_on_the_device(tmp_path):
    """`TorchScriptRunner(path, device="cuda")` evaluates the archive's network with the fused kernel (no TorchScript
    interpreter on the device), `EvalBatcher` packs four threads' requests into the engine's batch: both equal the
    engine's direct forward bit for bit, and stay within the fused kernel's bounds of the fp32 module."""
    import threading
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    torch.manual_seed(20260314)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    path = str(tmp_path / "b6c64.pt")
    torch.jit.trace(model, torch.zeros(2, 11, 6, 6)).save(path)
    z = load("g1_rules.npz")
    st = states(z, "s")
    planes = v0_core.states_to_model_input(*[torch.from_numpy(np.ascontiguousarray(np.asarray(st[f])[:96])).to(DEV)
                                             for f in ("board", "marks_black", "marks_white", "phase", "current_player")])
    runner = v0_core.TorchScriptRunner(path, DEV, "auto")
    assert (runner.device, runner.dtype) == (DEV, "auto") and runner.fused is not None and runner.module is None
    engine = v0_core.InferenceEngine(path, device=DEV, dtype="float16", batch_size=64)
    direct = engine.forward(planes[:64], 64)
    for a, b in zip(runner.forward(planes[:64]), direct):
        assert torch.equal(a, b)
    with torch.no_grad():
        want = model.to(DEV)(planes[:64])
    for k in range(3):
        assert float((direct[k].exp() - want[k].exp()).abs().max()) < 1e-4
    with pytest.raises(RuntimeError, match="float16 or float32"):
        v0_core.TorchScriptRunner(path, DEV, "bfloat16")
    batcher = v0_core.EvalBatcher(engine, 64, timeout_ms=5)
    chunks = [(0, 10), (10, 30), (30, 37), (37, 64), (64, 96)]
    out = [None] * len(chunks)

    def call(i):
        a, b = chunks[i]
        out[i] = batcher.forward(planes[a:b].cpu())                 # host inputs travel to the engine's device
    threads = [threading.Thread(target=call, args=(i,)) for i in range(len(chunks))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for (a, b), o in zip(chunks, out):
        ref = engine.forward(planes[a:b], b - a)
        for p, q in zip(o, ref):
            assert p.shape == q.shape and float((p - q).abs().max()) < 1e-5     # a sample's result does not depend on its batch neighbours
    stats = batcher.get_eval_stats()
    assert stats["eval_leaves"] == 96 and 2 <= stats["eval_calls"] <= 5
    batcher.shutdown()


def test_mcts_core_searches_through_an_eval_batcher_and_a_torchscript_runner(tmp_path):
    """`set_eval_batcher` / `set_torchscript_runner` (module.cpp:1196-1230): the same search as through the engine itself."""
    from liuzhou_amd import v0_core
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    torch.manual_seed(20260314)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    path = str(tmp_path / "b6c64.pt")
    torch.jit.trace(model, torch.zeros(2, 11, 6, 6)).save(path)
    engine = v0_core.InferenceEngine(path, device=DEV, dtype="float16", batch_size=8)
    batcher = v0_core.EvalBatcher(engine, 8, timeout_ms=0)
    runner = v0_core.TorchScriptRunner(path, DEV)
    st = states(load("g1_rules.npz"), "s")
    for i in (5, 400, 1200):
        seen = []
        for attach in (lambda c: c.set_inference_engine(engine), lambda c: c.set_eval_batcher(batcher),
                       lambda c: c.set_torchscript_runner(runner)):
            cfg = v0_core.MCTSConfig()
            cfg.num_simulations, cfg.device = 32, DEV
            core = v0_core.MCTSCore(cfg)
            attach(core)
            core.set_root_state(_state_like(st, i))
            core.run_simulations(32)
            seen.append([(s["action_index"], int(s["visit_count"])) for s in core.get_root_children_stats()])
        assert seen[0] == seen[1] == seen[2] and sum(v for _, v in seen[0]) in (31, 32)
    assert batcher.get_eval_stats()["eval_calls"] >= 32
    batcher.shutdown()
    with pytest.raises(RuntimeError, match="null"):
        v0_core.MCTSCore(v0_core.MCTSConfig()).set_eval_batcher(None)
