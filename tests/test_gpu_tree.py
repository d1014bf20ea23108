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
