"""GPU: the `MCTSConfig` / `MCTSCore` / `InferenceEngine` adapters (v0_core class surface)."""
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
    with pytest.raises(RuntimeError):
        core.set_torchscript_runner(object())
