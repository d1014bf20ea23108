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


@pytest.fixture(scope="module", params=["native", "python"])
def v0(request):
    """Both bindings of the C ABI: the compiled PyBind11 layer and the ctypes layer (tests/test_gpu_ops.py::V0Binding)."""
    return G.binding_or_skip(request.param)


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


def test_inplace_operators_refuse_wrong_dtypes_shapes_and_devices(v0):
    """The in-place operators hand the caller's storages to the kernels unconverted, so both bindings check what the other
    operators convert: an int32 `phase`, a short scalar tensor, codes that do not align with the slots."""
    t = list(G.to_dev(O.initial_states(4)))
    codes = torch.tensor([[1, 5, -1, -1], [1, 7, -1, -1]], dtype=torch.int32)
    slots = torch.tensor([0, 2])
    v0.batch_apply_moves_inplace(*t, codes, slots)                     # the well-formed call works
    assert t[10].tolist() == [1, 0, 1, 0]
    bad = list(t); bad[3] = t[3].to(torch.int32)
    with pytest.raises(RuntimeError, match="phase must be int64"):
        v0.batch_apply_moves_inplace(*bad, codes, slots)
    bad = list(t); bad[0] = t[0].to(torch.int64)
    with pytest.raises(RuntimeError, match="board must be int8"):
        v0.batch_apply_moves_inplace(*bad, codes, slots)
    bad = list(t); bad[6] = t[6][:3].clone()
    with pytest.raises(RuntimeError, match="does not hold 4 states"):
        v0.batch_apply_moves_inplace(*bad, codes, slots)
    with pytest.raises(RuntimeError, match=r"\[N, 4\]"):
        v0.batch_apply_moves_inplace(*t, codes, torch.tensor([0, 1, 2]))
    with pytest.raises(RuntimeError, match=r"\[N, 4\]"):
        v0.batch_apply_moves_inplace(*t, codes[:, :3], slots)
    plies, done = torch.zeros(4, dtype=torch.int64), torch.zeros(4, dtype=torch.bool)
    step = lambda st, pl=plies, dn=done: v0.self_play_step_inplace(
        *st, pl, dn, torch.tensor([1]), torch.tensor([[1, 9, -1, -1]], dtype=torch.int32), torch.tensor([False]),
        torch.tensor([True]), 64, 2.0)
    step(t)
    assert plies.tolist() == [0, 1, 0, 0]
    bad = list(t); bad[1] = t[1].to(torch.uint8)
    with pytest.raises(RuntimeError, match="marks_black must be bool"):
        step(bad)
    with pytest.raises(RuntimeError, match="one entry per state"):
        step(t, pl=torch.zeros(3, dtype=torch.int64))


def test_native_binding_is_built_and_active():
    """`__graft_entry__.build()` compiles csrc/v0_core_ext.cpp; the module-level operators are then the compiled ones."""
    from liuzhou_amd import v0_core
    from liuzhou_amd.build import ext_path
    import os
    assert os.path.exists(ext_path()), "liuzhou_amd/_v0_core_native*.so is missing: python -m liuzhou_amd.build"
    assert v0_core.active_binding() == "native", v0_core._native_error
    assert v0_core.encode_actions_fast is v0_core.binding("native").encode_actions_fast
    assert set(vars(v0_core.binding("python"))) == set(vars(v0_core.binding("native")))


def _rand_pack(seed, R=7, M=9, B=12, T=220):
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack([torch.randperm(T, generator=g)[:M] for _ in range(R)])
    codes = torch.randint(-1, 36, (R, M, 4), generator=g, dtype=torch.int32)
    valid = torch.rand(R, M, generator=g) < 0.7
    valid[:, 0] = True
    pol = torch.rand(R, M, generator=g) * valid
    pol = pol / pol.sum(1, keepdim=True)
    picks = torch.randint(0, M, (R,), generator=g)
    roots = torch.randperm(B, generator=g)[:R]
    return idx, codes, valid, pol, picks, roots, B, T


def test_defined_but_unused_reference_ops(v0):
    """`root_sparse_writeback` (module.cpp:365-439), `postprocess_value_head`, `apply_temperature_scaling`
    (v0/src/net/encoding.cpp:81-113): the reference defines them, its v1 path does not call them; here they exist on both
    bindings, agree with each other, with a direct restatement, and -- in the build container -- with the reference's own
    compiled module (oracle/_ref)."""
    idx, codes, valid, pol, picks, roots, B, T = _rand_pack(3)
    out = v0.root_sparse_writeback(idx, codes, valid, pol, picks, roots, B, T)
    policy_dense, chosen_idx, chosen_codes, chosen_valid = out
    assert policy_dense.shape == (B, T) and chosen_idx.dtype == torch.int64 and chosen_codes.dtype == torch.int32
    for r in range(idx.shape[0]):
        b = int(roots[r])
        want = torch.zeros(T)
        want[idx[r]] = (pol[r] * valid[r]).float()
        assert torch.allclose(policy_dense[b], want) and bool(chosen_valid[b])
        assert int(chosen_idx[b]) == int(idx[r, picks[r]]) and torch.equal(chosen_codes[b], codes[r, picks[r]])
    rest = torch.ones(B, dtype=torch.bool); rest[roots] = False
    assert bool((chosen_idx[rest] == -1).all()) and bool((chosen_codes[rest] == -1).all()) and not bool(chosen_valid[rest].any())
    assert float(policy_dense[rest].abs().sum()) == 0.0
    g = torch.Generator().manual_seed(4)
    wdl, scalar = torch.randn(6, 3, generator=g), torch.randn(6, 1, generator=g)
    p = torch.softmax(wdl, -1)
    assert torch.allclose(v0.postprocess_value_head(wdl), p[:, 0] - p[:, 2]) and torch.allclose(v0.postprocess_value_head(scalar), torch.tanh(scalar))
    probs = torch.rand(5, 8, generator=g); probs[:, 2] = 0; probs[4] = 0
    got = v0.apply_temperature_scaling(probs, 0.5)
    want = probs.pow(2.0); want = torch.where(want.sum(1, keepdim=True) > 0, want / want.sum(1, keepdim=True), torch.zeros_like(want))
    assert torch.allclose(got, want) and torch.equal(v0.apply_temperature_scaling(probs, 0.0), probs)
    assert torch.allclose(v0.apply_temperature_scaling(probs.t().contiguous(), 0.5, 0), want.t())
    with pytest.raises(RuntimeError):
        v0.root_sparse_writeback(idx, codes[:, :, :3], valid, pol, picks, roots, B, T)
    import glob, importlib.util, os
    ref = glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "v0_core*.so"))
    if ref:
        spec = importlib.util.spec_from_file_location("v0_core", ref[0])
        R = importlib.util.module_from_spec(spec); spec.loader.exec_module(R)
        for a, b in zip(out, R.root_sparse_writeback(idx, codes, valid, pol, picks, roots, B, T)):
            assert a.dtype == b.dtype and torch.equal(a, b)
        assert torch.equal(v0.postprocess_value_head(wdl), R.postprocess_value_head(wdl))
        assert torch.equal(v0.apply_temperature_scaling(probs, 0.5), R.apply_temperature_scaling(probs, 0.5, -1))


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
