"""GPU: the reference's own hand-built unit cases for the search's read-out, replayed on the device engine.

tests/v1/test_portable_mcts.py builds small visit / value / prior vectors by hand and checks the pick and the policy
target; here the same vectors are written straight into a one-game tree arena (root edge records) and read out by
`lz_tree_finish`, the kernel the production path uses.  (Visit-count semantics -- sign flips only when the mover changes,
deep negative evidence, fixed q after the first visit, no-legal = loss, subtree reuse, fresh noise on a kept root -- are
pinned by the recorded reference searches g5 / g10 / g13 and the oracle replays of tests/test_gpu_tree.py.)"""
import numpy as np
import pytest
import torch

from oracle import lz_oracle as O
from tests.tree_parity import EDGE_DT, NODE_DT, to_gpu_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine_with_root_edges(actions, visits, q_root_side, priors):
    """A one-game arena whose root (the initial position, black to move) has the given children; q is given from the
    root mover's side and the children are white-to-move positions, so W = -q * N in the child mover's perspective."""
    from liuzhou_amd.tree_engine import TreeEngine
    eng = TreeEngine(1, 64, DEV, 1.0)
    eng.set_roots(to_gpu_batch(O.initial_states(1), DEV))
    eng.begin()
    n = len(actions)
    edges = np.zeros(n, EDGE_DT)
    edges["act"] = actions
    edges["P"] = priors
    edges["n_info"] = np.asarray(visits, np.uint32) | (np.uint32(1) << 24)          # info bit 0: child mover is white
    edges["W"] = -np.asarray(q_root_side, np.float64) * np.asarray(visits, np.float64)
    edges["child"] = -1
    node = eng.buf["nodes"].view(1, eng.node_cap, 6)[0, :1].cpu().numpy().view(NODE_DT).copy()
    # the root's run at the start of chunk 0 of the edge pool, booked as the game's first chunk
    node["edge_begin"], node["nedges"], node["parent"] = 0, n, -1
    eng.buf["nodes"].view(1, eng.node_cap, 6)[0, :1] = torch.from_numpy(node.view(np.int64).reshape(1, 6)).to(DEV)
    eng.buf["edges"][:n] = torch.from_numpy(edges.view(np.int64).reshape(n, 4)).to(DEV)
    eng.buf["n_edges"].fill_(n)
    eng.buf["chunk_list"][0] = 0
    eng.buf["n_chunks"].fill_(1)
    eng.buf["free_chunks"].copy_(torch.roll(torch.arange(eng.pool_chunks, dtype=torch.int32, device=DEV), -1))
    eng.buf["pool_top"].fill_(eng.pool_chunks - 1)
    eng.buf["root_visits"].fill_(int(np.sum(visits)))
    eng.buf["root_w"].fill_(float(np.sum(np.asarray(q_root_side) * np.asarray(visits))))
    eng.buf["root_terminal"].zero_()
    return eng


def test_deterministic_action_breaks_visit_ties_by_q_then_prior_then_index():
    """test_portable_mcts.py:349-357 (deterministic_action_from_search, portable_mcts.py:208-261): most visits, then Q
    within 1e-6, then prior within 1e-8, then the lowest action index.  The reference's vectors: visits 4/4/4/4, values
    .1/.3/.3/.3, priors .9/.2/.4/.4 -> the third action."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    eng = _engine_with_root_edges([0, 1, 2, 3], [4, 4, 4, 4], [0.1, 0.3, 0.3, 0.3], [0.9, 0.2, 0.4, 0.4])
    eng.finish(torch.ones(1, device=DEV), None, sample_moves=False)
    assert int(eng.chosen_index.item()) == 2
    assert eng.chosen_code[0].tolist() == [1, 2, -1, -1]                # placement at cell 2
    # a strictly larger visit count wins whatever Q and P say; exact ties fall to the lowest index
    eng = _engine_with_root_edges([5, 9, 12], [3, 7, 7], [0.9, -0.5, -0.5], [0.8, 0.1, 0.1])
    eng.finish(torch.ones(1, device=DEV), None, sample_moves=False)
    assert int(eng.chosen_index.item()) == 9


def test_policy_target_prior_pseudocount_preserves_all_legal_actions():
    """test_portable_mcts.py:320-346 (policy_from_visits_and_priors, portable_mcts.py:150-205): visits 8/0/0/0 with priors
    .4/.3/.2/.1 -> one-hot without the pseudocount, (visits + priors) / sum with pseudocount 1 at temperature 1."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    acts, visits, priors = [0, 1, 2, 3], [8, 0, 0, 0], [0.4, 0.3, 0.2, 0.1]
    eng = _engine_with_root_edges(acts, visits, [0.0] * 4, priors)
    t1 = torch.ones(1, device=DEV)
    eng.finish(t1, None, sample_moves=False)
    assert eng.policy_dense[0, :4].tolist() == [1.0, 0.0, 0.0, 0.0] and float(eng.policy_dense.sum()) == 1.0
    eng.finish(t1, None, target_temperatures=t1, prior_pseudocount=1.0, sample_moves=False)
    got = eng.policy_dense[0, :4].cpu().numpy()
    want = (np.asarray(visits, np.float32) + np.asarray(priors, np.float32))
    want = want / want.sum()
    assert (got > 0).all() and abs(float(got.sum()) - 1.0) <= 1e-6
    np.testing.assert_allclose(got, want, atol=1e-6, rtol=0)
    assert float(eng.policy_dense[0, 4:].abs().sum()) == 0.0            # nothing outside the root's children
    assert int(eng.chosen_index.item()) == 0                             # the target options do not change the move


def test_policy_target_temperature_does_not_change_action_selection():
    """test_portable_mcts.py:373-402: the training target may be sharpened / flattened by its own temperature while the
    move is picked from the selection policy."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    eng = _engine_with_root_edges([3, 4, 7], [10, 30, 20], [0.0, 0.1, 0.2], [0.3, 0.3, 0.4])
    t1 = torch.ones(1, device=DEV)
    picks, targets = [], []
    for tt in (None, torch.full((1,), 0.25, device=DEV), torch.full((1,), 4.0, device=DEV)):
        eng.finish(t1, None, target_temperatures=tt, sample_moves=False)
        picks.append(int(eng.chosen_index.item()))
        targets.append(eng.policy_dense[0, [3, 4, 7]].cpu().numpy().copy())
    assert picks == [4, 4, 4]
    np.testing.assert_allclose(targets[0], np.array([10, 30, 20]) / 60.0, atol=1e-6)
    v = np.array([10.0, 30.0, 20.0])
    for tt, got in ((0.25, targets[1]), (4.0, targets[2])):
        w = v ** (1.0 / tt)
        np.testing.assert_allclose(got, w / w.sum(), atol=1e-5)
