"""CPU: the oracle's variant-R self-play loop reproduces the reference v1 runner's trace (g8)."""
import numpy as np
import torch

from oracle import selfplay_oracle as SO
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from tests.golden_utils import load


def test_oracle_root_selfplay_matches_reference_trace():
    z = load("g8_selfplay.npz")
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval()
    tensors, stats = SO.self_play_root(model, num_games=4, sims=32, temperature_init=1.0, temperature_final=0.1,
                                       temperature_threshold=10, c=1.0, soft_k=2.0, max_game_plies=512)
    n = int(z["num_positions"])
    assert stats["num_positions"] == n
    want_states = np.unpackbits(z["state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(tensors["state_tensors"], want_states)
    assert np.array_equal(tensors["legal_masks"], np.unpackbits(z["legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(tensors["policy_targets"], z["policy_targets"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(tensors["value_targets"], z["value_targets"])
    np.testing.assert_allclose(tensors["soft_value_targets"], z["soft_value_targets"], atol=1e-6, rtol=0)
    assert (stats["black_wins"], stats["white_wins"], stats["draws"]) == (int(z["black_wins"]), int(z["white_wins"]), int(z["draws"]))


def _check_sparse(tensors, stats, z, tag):
    n = z[f"{tag}_policy_targets"].shape[0]
    assert stats["num_positions"] == n
    want_states = np.unpackbits(z[f"{tag}_state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(np.asarray(tensors["state_tensors"]), want_states)
    assert np.array_equal(np.asarray(tensors["legal_masks"]),
                          np.unpackbits(z[f"{tag}_legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(np.asarray(tensors["policy_targets"]), z[f"{tag}_policy_targets"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(np.asarray(tensors["value_targets"]), z[f"{tag}_value_targets"])
    np.testing.assert_allclose(np.asarray(tensors["soft_value_targets"]), z[f"{tag}_soft_value_targets"], atol=1e-6, rtol=0)


def test_oracle_root_selfplay_with_topk_lookahead_matches_reference_traces():
    """sparse_ply = 2 and 3 (the reference's experimental multi-ply refinement, mcts_gpu.py:976-1160): g12."""
    z = load("g12_sparse_selfplay.npz")
    for tag in ("p2k4", "p3k3"):
        ply, top_k, games, sims, max_plies = (int(x) for x in z[f"{tag}_config"])
        torch.manual_seed(7)
        model = ChessNet(**MODEL_CONFIGS["tiny"]).eval()
        tensors, stats = SO.self_play_root(model, num_games=games, sims=sims, temperature_init=1.0,
                                           temperature_final=0.1, temperature_threshold=10, c=1.0, soft_k=2.0,
                                           max_game_plies=max_plies, sparse_ply=ply, sparse_top_k=top_k)
        _check_sparse(tensors, stats, z, tag)
        assert [stats["black_wins"], stats["white_wins"], stats["draws"]] == z[f"{tag}_outcome"].tolist()


def test_oracle_legacy_wave_search_matches_reference_visit_counts():
    """src/mcts.py with batch_K = 16 / 4 (leaves collected in waves, no virtual loss), three consecutive searches
    with advance_root in between: the oracle's wave search reproduces the reference's root visit counts (g13)."""
    from oracle import lz_oracle as O
    from tests.golden_utils import states
    z = load("g13_legacy_waves.npz")
    roots = states(z, "r")
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval()
    evaluate = SO.make_net_evaluator(model)
    n = z["case_root"].shape[0]
    assert n >= 60 and int(z["case_k"].max()) == 16
    for ci in range(n):
        cur = O.state_from_batch(roots, int(z["case_root"][ci]))
        sims, k = int(z["case_sims"][ci]), int(z["case_k"][ci])
        tree = O.OracleTree(cur, 1.0)
        for mv in range(int(z["case_moves"][ci])):
            SO.tree_search_waves(evaluate, [tree], sims, k)
            idx, vis, _, _, _ = tree.root_children()
            got = np.zeros(220, np.int32); got[idx] = vis
            assert np.array_equal(got, z["case_visits"][ci, mv]), (ci, mv, sims, k)
            pick = int(np.flatnonzero(got == got.max())[0])
            cur = O.apply_index(cur, pick)
            if not tree.advance(pick):
                tree = O.OracleTree(cur, 1.0)


def _tree_trace(z, tag, **kw):
    """The recorded games turn on the last bits of the tiny net's near-uniform priors, and a host with another vector
    ISA rounds the convolutions differently: the oracle replays the network outputs the reference run itself produced
    (recorded at PortableMCTS.evaluate_states by oracle/gen_golden.py) instead of re-evaluating the module."""
    games, sims, max_plies = (int(x) for x in z[f"{tag}_config"])
    table = SO.make_table_evaluator(z[f"{tag}_eval_planes"], z[f"{tag}_eval_priors"], z[f"{tag}_eval_values"])
    out = SO.self_play_tree(None, num_games=games, sims=sims, temperature_init=1.0, temperature_final=0.1,
                            temperature_threshold=10, c=1.0, soft_k=2.0, max_game_plies=max_plies,
                            concurrent_games=games, reuse_tree=True, collect=True, evaluate=table, **kw)
    t = out["tensors"]
    n = z[f"{tag}_policy_targets"].shape[0]
    assert out["num_positions"] == n
    want_states = np.unpackbits(z[f"{tag}_state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(t["state_tensors"], want_states)
    assert np.array_equal(t["legal_masks"], np.unpackbits(z[f"{tag}_legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(t["policy_targets"], z[f"{tag}_policy_targets"], atol=1e-6, rtol=0)
    np.testing.assert_array_equal(t["value_targets"], z[f"{tag}_value_targets"])
    np.testing.assert_allclose(t["soft_value_targets"], z[f"{tag}_soft_value_targets"], atol=1e-6, rtol=0)
    assert [out["black_wins"], out["white_wins"], out["draws"]] == [int(x) for x in z[f"{tag}_outcome"]]


def test_oracle_tree_selfplay_with_subtree_reuse_matches_reference_trace():
    """g10/a: the reference portable runner keeps the played child's subtree on every move (advance_root)."""
    _tree_trace(load("g10_tree_selfplay.npz"), "a")


def test_oracle_tree_selfplay_policy_target_options_match_reference_trace():
    """g10/b: policy_target_temperature / policy_target_prior_pseudocount (portable_mcts.py:690-700)."""
    _tree_trace(load("g10_tree_selfplay.npz"), "b", policy_target_temperature=1.0, policy_target_prior_pseudocount=0.5)


def test_oracle_tree_selfplay_with_the_module_itself_on_this_host():
    """Same trace with the tiny module evaluated here (the fixtures were generated in this container; skipped where the
    host's fp32 convolutions round differently, which the recorded-evaluation test above is immune to)."""
    import pytest
    z = load("g10_tree_selfplay.npz")
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval()
    games, sims, max_plies = (int(x) for x in z["a_config"])
    out = SO.self_play_tree(model, num_games=games, sims=sims, temperature_init=1.0, temperature_final=0.1,
                            temperature_threshold=10, c=1.0, soft_k=2.0, max_game_plies=max_plies,
                            concurrent_games=games, reuse_tree=True, collect=True)
    n = z["a_policy_targets"].shape[0]
    want = np.unpackbits(z["a_state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    if out["num_positions"] != n or not np.array_equal(out["tensors"]["state_tensors"], want):
        pytest.skip("this host's fp32 convolutions round differently from the recording host's")
    np.testing.assert_allclose(out["tensors"]["policy_targets"], z["a_policy_targets"], atol=1e-6, rtol=0)
