"""CPU: the `v0_core` operators on CPU tensors -- the host build of the C ABI (libliuzhou_host.so, csrc/lz_host.cpp), which
the module dispatches to by tensor device like the reference extension (v0/src/game/fast_legal_mask.cpp:453) -- against the
same golden vectors and oracle comparisons as the HIP kernels (the test bodies ARE those of tests/test_gpu_ops.py, run with
the device switched to "cpu"), plus the reference's CPU error convention: an illegal action raises."""
import numpy as np
import pytest
import torch

from oracle import lz_oracle as O
from tests import test_gpu_ops as G
from tests.golden_utils import load, states, FIELDS


@pytest.fixture(scope="module")
def v0():
    from liuzhou_amd import v0_core
    return v0_core


@pytest.fixture(autouse=True)
def _on_cpu(monkeypatch):
    monkeypatch.setattr(G, "DEV", "cpu")


test_encode_actions_reachable_bit_exact = G.test_encode_actions_reachable_bit_exact
test_encode_actions_garbage_bit_exact = G.test_encode_actions_garbage_bit_exact
test_encode_actions_large_random_vs_oracle = G.test_encode_actions_large_random_vs_oracle
test_encode_actions_empty_batch = G.test_encode_actions_empty_batch
test_encode_actions_reference_suite_10000_garbage_states = G.test_encode_actions_reference_suite_10000_garbage_states
test_batch_apply_moves_reference_suite_10000_micro_positions = G.test_batch_apply_moves_reference_suite_10000_micro_positions
test_rules_reference_suite_5000_playout_states = G.test_rules_reference_suite_5000_playout_states
test_batch_apply_moves_all_transitions_bit_exact = G.test_batch_apply_moves_all_transitions_bit_exact
test_batch_apply_moves_inplace = G.test_batch_apply_moves_inplace
test_states_to_model_input_exact = G.test_states_to_model_input_exact
test_project_policy_logits = G.test_project_policy_logits
test_root_pack_sparse_actions = G.test_root_pack_sparse_actions
test_root_puct_visit_counts_bit_exact = G.test_root_puct_visit_counts_bit_exact
test_root_puct_random_vs_oracle = G.test_root_puct_random_vs_oracle
test_root_finalize_from_visits = G.test_root_finalize_from_visits
test_self_play_step_inplace = G.test_self_play_step_inplace
test_finalize_trajectory_inplace = G.test_finalize_trajectory_inplace


def test_illegal_action_raises_on_cpu_tensors(v0):
    """fast_apply_moves.cpp:264-470: the reference's CPU path TORCH_CHECKs every action (the GPU path is a silent no-op)."""
    st = O.initial_states(2)
    t = G.to_dev(st)
    legal = torch.tensor([[1, 5, -1, -1], [1, 7, -1, -1]], dtype=torch.int32)
    out = v0.batch_apply_moves(*t, legal, torch.tensor([0, 1]))
    assert int(out[0].ne(0).sum()) == 2 and out[10].tolist() == [1, 1]
    bad = torch.tensor([[2, 5, 0, -1]], dtype=torch.int32)             # a movement during placement
    with pytest.raises(RuntimeError, match="illegal action"):
        v0.batch_apply_moves(*t, bad, torch.tensor([0]))


def test_host_library_exports_the_operator_subset_with_header_signatures():
    from liuzhou_amd import _lib
    H = _lib.host_lib()
    for name in _lib.HOST_SYMBOLS:
        assert getattr(H, name).argtypes == _lib.DECLS[name][1]
    assert b"host" in H.lz_version()
    # the engines have no host build: CPU devices are refused
    from liuzhou_amd.tree_engine import TreeEngine
    with pytest.raises(RuntimeError, match="HIP device"):
        TreeEngine(4, 8, "cpu")
