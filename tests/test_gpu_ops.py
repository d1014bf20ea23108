"""GPU parity: every v0_core operator (HIP kernels through the C ABI) against the oracle and the
reference's golden vectors.  Integer / byte / index outputs bit-exact; float outputs <= 1e-5
(the tolerance north_star states for policy/value tensors)."""
import numpy as np
import pytest
import torch

from oracle import lz_oracle as O
from tests.golden_utils import load, states, unpack_mask, states_equal, FIELDS, row_hash64, group_hash64, state_rows

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


class V0Binding:
    """`v0_core` with its operators taken from ONE binding of the C ABI: "native" = the compiled PyBind11 layer
    (csrc/v0_core_ext.cpp), "python" = the ctypes layer; everything else comes from the module."""

    def __init__(self, kind: str) -> None:
        from liuzhou_amd import v0_core
        self._mod, self._ops, self.kind = v0_core, vars(v0_core.binding(kind)), kind

    def __getattr__(self, name):
        ops = object.__getattribute__(self, "_ops")
        return ops[name] if name in ops else getattr(object.__getattribute__(self, "_mod"), name)


def binding_or_skip(kind: str) -> V0Binding:
    from liuzhou_amd import v0_core
    if kind == "native" and v0_core._native is None:
        pytest.skip(f"compiled v0_core layer unavailable: {v0_core._native_error}")
    return V0Binding(kind)


@pytest.fixture(scope="module", params=["native", "python"])
def v0(request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return binding_or_skip(request.param)


def to_dev(st):
    out = []
    for f in FIELDS:
        a = np.ascontiguousarray(st[f])
        if f == "board":
            out.append(torch.from_numpy(a.astype(np.int8)).to(DEV))
        elif f.startswith("marks"):
            out.append(torch.from_numpy(a.astype(bool)).to(DEV))
        else:
            out.append(torch.from_numpy(a.astype(np.int64)).to(DEV))
    return out


def from_dev(ts):
    return {f: t.cpu().numpy() for f, t in zip(FIELDS, ts)}


def test_encode_actions_reachable_bit_exact(v0):
    z = load("g1_rules.npz")
    t = to_dev(states(z, "s"))
    mask, meta = v0.encode_actions_fast(*t[:10], 36, 144, 36, 4)
    assert mask.dtype == torch.bool and meta.dtype == torch.int32
    assert np.array_equal(mask.cpu().numpy(), unpack_mask(z["legal_mask"], 220))
    assert np.array_equal(meta.cpu().numpy(), z["metadata"].astype(np.int32))


@pytest.mark.parametrize("aux,key", [(1, "t217"), (4, "t220")])
def test_encode_actions_garbage_bit_exact(v0, aux, key):
    """tests/v0/cuda/test_fast_legal_mask_cuda.py:179-228 equivalent: synthetic unreachable states."""
    z = load("g3_garbage.npz")
    t = to_dev(states(z, "s"))
    mask, meta = v0.encode_actions_fast(*t[:10], 36, 144, 36, aux)
    assert np.array_equal(mask.cpu().numpy(), unpack_mask(z[f"mask_{key}"], 216 + aux))
    assert np.array_equal(meta.cpu().numpy(), z[f"meta_{key}"].astype(np.int32))


def test_encode_actions_large_random_vs_oracle(v0):
    rng = np.random.default_rng(0xF00D)
    n = 10000
    st = O.empty_states(n)
    st["board"] = rng.integers(-1, 2, (n, 6, 6)).astype(np.int8)
    dense = rng.random(n) < 0.4
    st["board"][dense] = np.where(rng.random((int(dense.sum()), 6, 6)) < 0.88, 1, -1).astype(np.int8) * rng.choice([-1, 1], (int(dense.sum()), 1, 1)).astype(np.int8)
    st["marks_black"] = rng.random((n, 6, 6)) < 0.2
    st["marks_white"] = rng.random((n, 6, 6)) < 0.2
    st["phase"] = rng.integers(1, 8, n)
    st["current_player"] = rng.choice([-1, 1], n)
    st["pending_marks_remaining"] = rng.integers(0, 3, n)
    st["pending_captures_remaining"] = rng.integers(0, 3, n)
    st["forced_removals_done"] = rng.integers(0, 3, n)
    want_mask, want_meta = O.encode_actions(st)
    t = to_dev(st)
    mask, meta = v0.encode_actions_fast(*t[:10], 36, 144, 36, 4)
    assert np.array_equal(mask.cpu().numpy(), want_mask)
    assert np.array_equal(meta.cpu().numpy(), want_meta)


# ---- reference-scale suites (oracle/gen_golden_large.py): the sizes, generators and seeds of the reference's own tests ----
@pytest.mark.parametrize("aux,key", [(1, "t217"), (4, "t220")])
def test_encode_actions_reference_suite_10000_garbage_states(v0, aux, key):
    """The 10 000 synthetic states of tests/v0/cuda/test_fast_legal_mask_cuda.py:74-118,179-228 (its generator, its seed
    0xF00DCAFE) against the reference CPU operator's outputs: mask bit-exact, metadata by 64-bit row hash."""
    z = load("g16_garbage_large.npz")
    t = to_dev(states(z, "s"))
    assert t[0].shape[0] == 10000
    mask, meta = v0.encode_actions_fast(*t[:10], 36, 144, 36, aux)
    assert np.array_equal(mask.cpu().numpy(), unpack_mask(z[f"mask_{key}"], 216 + aux))
    bad = np.flatnonzero(row_hash64(meta.cpu().numpy()) != z[f"meta_hash_{key}"])
    assert bad.size == 0, f"metadata rows differ: {bad[:10]}"
    if aux == 4:
        planes = v0.states_to_model_input(*t[:5])
        assert np.array_equal(row_hash64(planes.cpu().numpy().astype(np.int8)), z["model_input_hash"])


def test_batch_apply_moves_reference_suite_10000_micro_positions(v0):
    """The 10 000 per-kind micro-positions of tests/v0/cuda/test_fast_apply_moves_cuda.py:127-247,343-373 (its generator,
    its seed 0xA11CEB0B): all 12 output tensors equal the reference CPU operator's."""
    z = load("g17_apply_micro.npz")
    t = to_dev(states(z, "s"))
    n = t[0].shape[0]
    assert n == 10000
    codes = torch.from_numpy(z["codes"].astype(np.int32)).to(DEV)
    parents = torch.arange(n, dtype=torch.int64, device=DEV)
    out = v0.batch_apply_moves(*t, codes, parents)
    ok, field = states_equal(from_dev(out), states(z, "c"))
    assert ok, field


def test_rules_reference_suite_5000_playout_states(v0):
    """>= 5 000 reachable states of src/ random playouts (seed 0x7777, tests/v0/test_actions.py:115-159 scale): legal
    mask bit-exact, metadata by row hash, and EVERY child transition (per-state hash of the children in ascending action
    order) against the reference's Python rule engine."""
    z = load("g15_rules_large.npz")
    st = states(z, "s")
    t = to_dev(st)
    n = t[0].shape[0]
    assert n >= 5000
    mask, meta = v0.encode_actions_fast(*t[:10], 36, 144, 36, 4)
    want = unpack_mask(z["legal_mask"], 220)
    assert np.array_equal(mask.cpu().numpy(), want)
    meta_h = meta.cpu().numpy()
    assert np.array_equal(row_hash64(meta_h), z["metadata_hash"])
    parents, actions = np.nonzero(want)                    # row-major: children grouped by parent, ascending action
    assert parents.size == int(z["num_children"])
    codes = torch.from_numpy(np.ascontiguousarray(meta_h[parents, actions])).to(DEV)
    out = v0.batch_apply_moves(*t, codes, torch.from_numpy(parents.astype(np.int64)).to(DEV))
    got = group_hash64(row_hash64(state_rows(from_dev(out))), parents, n)
    bad = np.flatnonzero(got != z["children_hash"])
    assert bad.size == 0, f"children of states {bad[:10]} differ"


def test_encode_actions_empty_batch(v0):
    t = to_dev(O.empty_states(0))
    mask, meta = v0.encode_actions_fast(*t[:10], 36, 144, 36, 4)
    assert tuple(mask.shape) == (0, 220) and tuple(meta.shape) == (0, 220, 4)


def test_batch_apply_moves_all_transitions_bit_exact(v0):
    z = load("g1_rules.npz")
    t = to_dev(states(z, "s"))
    parents = torch.from_numpy(z["child_parent"].astype(np.int64)).to(DEV)
    codes = torch.from_numpy(z["metadata"].astype(np.int32)[z["child_parent"], z["child_action"].astype(np.int64)]).to(DEV)
    out = v0.batch_apply_moves(*t, codes, parents)
    assert out[1].dtype == torch.bool and out[0].dtype == torch.int8 and out[3].dtype == torch.int64
    ok, field = states_equal(from_dev(out), states(z, "c"))
    assert ok, field


def test_batch_apply_moves_illegal_actions_noop_semantics(v0):
    z = load("g1_rules.npz")
    st = states(z, "s")
    B = st["board"].shape[0]
    rng = np.random.default_rng(5)
    N = 50000
    parents = rng.integers(0, B, N).astype(np.int64)
    codes = np.stack([rng.integers(0, 10, N), rng.integers(-1, 37, N), rng.integers(-1, 5, N),
                      rng.integers(-1, 36, N)], axis=1).astype(np.int32)
    want = O.apply_moves(st, codes, parents, strict=False)
    out = v0.batch_apply_moves(*to_dev(st), torch.from_numpy(codes).to(DEV), torch.from_numpy(parents).to(DEV))
    ok, field = states_equal(from_dev(out), want)
    assert ok, field


def test_batch_apply_moves_inplace(v0):
    z = load("g1_rules.npz")
    st = states(z, "s")
    B = st["board"].shape[0]
    mask = unpack_mask(z["legal_mask"], 220)
    meta = z["metadata"].astype(np.int32)
    first = mask.argmax(1)
    codes = meta[np.arange(B), first]
    slots = np.arange(B, dtype=np.int64)[::2].copy()
    want = O.apply_moves(st, codes[slots], slots, strict=True)
    t = to_dev(st)
    v0.batch_apply_moves_inplace(*t, torch.from_numpy(codes[slots]).to(DEV), torch.from_numpy(slots).to(DEV))
    got = from_dev(t)
    ok, field = states_equal({f: got[f][slots] for f in FIELDS}, want)
    assert ok, field
    untouched = np.arange(B)[1::2]
    ok, field = states_equal({f: got[f][untouched] for f in FIELDS}, {f: np.asarray(st[f])[untouched] for f in FIELDS})
    assert ok, field


def test_states_to_model_input_exact(v0):
    for name in ("g2_edges.npz", "g3_garbage.npz"):
        z = load(name)
        t = to_dev(states(z, "s"))
        x = v0.states_to_model_input(t[0], t[1], t[2], t[3], t[4])
        assert x.dtype == torch.float32 and tuple(x.shape[1:]) == (11, 6, 6)
        assert np.array_equal(x.cpu().numpy(), z["model_input"].astype(np.float32)), name


def test_project_policy_logits(v0):
    z = load("g4_project.npz")
    mask = torch.from_numpy(unpack_mask(z["mask"], 220)).to(DEV)
    probs, ml = v0.project_policy_logits_fast(*(torch.from_numpy(z[k]).to(DEV) for k in ("lp1", "lp2", "lpmc")),
                                              mask, 36, 144, 36, 4)
    np.testing.assert_allclose(probs.cpu().numpy(), z["probs"], atol=1e-6, rtol=0)   # tolerance: <= 1e-5 required
    got, want = ml.cpu().numpy(), z["masked_logits"]
    assert np.array_equal(np.isneginf(got), np.isneginf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], atol=1e-6, rtol=0)
    # larger random batch against the oracle
    rng = np.random.default_rng(3)
    B = 4096
    lp = [torch.log_softmax(torch.from_numpy(rng.normal(size=(B, 36)).astype(np.float32)) * 2, 1) for _ in range(3)]
    m = rng.random((B, 220)) < 0.12
    m[:, 217:] = False
    wp, wl = O.project_policy(lp[0].numpy(), lp[1].numpy(), lp[2].numpy(), m)
    probs, ml = v0.project_policy_logits_fast(*(x.to(DEV) for x in lp), torch.from_numpy(m).to(DEV), 36, 144, 36, 4)
    np.testing.assert_allclose(probs.cpu().numpy(), wp, atol=1e-6, rtol=0)


def test_root_pack_sparse_actions(v0):
    z = load("g7_ops.npz")
    mask = torch.from_numpy(unpack_mask(z["mask"], 220)).to(DEV)
    probs = torch.from_numpy(z["probs"]).to(DEV)
    meta = torch.from_numpy(z["meta"].astype(np.int32)).to(DEV)
    pack = v0.root_pack_sparse_actions(mask, probs, meta)
    names = ["terminal_mask", "valid_root_indices", "counts", "valid_mask", "legal_index_mat", "priors_mat",
             "action_code_mat", "pack_flat_idx", "action_codes_all", "parent_indices_all"]
    for n, got in zip(names, pack):
        want = z[f"pack_{n}"]
        g = got.cpu().numpy()
        assert g.dtype == want.dtype, (n, g.dtype, want.dtype)
        if g.dtype == np.float32:
            np.testing.assert_allclose(g, want, atol=1e-6, rtol=0, err_msg=n)
        else:
            assert np.array_equal(g, want), n


def test_root_puct_visit_counts_bit_exact(v0):
    z = load("g6_root_puct.npz")
    p, lv, vm = (torch.from_numpy(z[k]).to(DEV) for k in ("priors", "leaf", "valid"))
    for sims in (1, 16, 200, 1024):
        v, vs, rv = v0.root_puct_allocate_visits(p, lv, vm, sims, 1.0)
        assert np.array_equal(v.cpu().numpy(), z[f"visits_{sims}"]), sims
        np.testing.assert_allclose(vs.cpu().numpy(), z[f"value_sum_{sims}"], atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(rv.cpu().numpy(), z[f"root_{sims}"], atol=1e-5, rtol=0)
    v, _, _ = v0.root_puct_allocate_visits(p, lv, vm, 64, 2.5)
    assert np.array_equal(v.cpu().numpy(), z["visits_64_c25"])


@pytest.mark.parametrize("A,sims", [(36, 1024), (72, 8192), (130, 3000), (25, 65536)])
def test_root_puct_table_kernel_equals_division_kernel(v0, A, sims, monkeypatch):
    """The table-driven pulls (sqrt table, x / integer as a double product: csrc/lz_ops.hip::puct_pulls) against the plain
    IEEE-division kernel (LZ_ROOT_PUCT_DIV=1), bit for bit on visits AND value sums, over ordinary rows and rows built to
    hurt: equal scores, zero / negative / huge / NaN / -inf entries, and magnitudes below 2^-100 (those rows must take
    the division branch of the new kernel)."""
    rng = np.random.default_rng(1000 + A)
    R = 96 if sims > 10000 else 384
    valid = rng.random((R, A)) < 0.6
    valid[:, rng.integers(0, A)] = True
    pri = (rng.random((R, A)) ** 3 * valid).astype(np.float32)
    pri /= np.maximum(pri.sum(1, keepdims=True), 1e-8)
    leaf = ((rng.random((R, A)) * 2 - 1) * valid).astype(np.float32)
    pri[0] = 1.0 / A; leaf[0] = 0.25; valid[0] = True                       # every score ties at every pull
    leaf[1] = 0.0; pri[2] = 0.0                                              # q == 0 everywhere / u == 0 everywhere
    leaf[3, :3] = [-1.0, 1.0, -0.0]; pri[3, :3] = [0.5, 1e-30, 0.25]
    pri[4, 0] = np.float32(1e-40); leaf[4, 1] = np.float32(3e-39)           # denormal inputs: division branch
    pri[5, 0] = np.float32(2.0 ** -110); leaf[5, 2] = np.float32(-2.0 ** -120)
    pri[6, 0] = np.nan; pri[6, 1] = np.inf; leaf[7, 0] = -np.inf; leaf[8, 0] = np.nan
    pri[9] = 3e38; leaf[10] = -3e38
    valid[4:11, :3] = True
    args = [torch.from_numpy(x).to(DEV) for x in (pri, leaf, valid)]
    monkeypatch.delenv("LZ_ROOT_PUCT_DIV", raising=False)
    v_t, vs_t, rv_t = (t.cpu().numpy() for t in v0.root_puct_allocate_visits(*args, sims, 1.5))
    monkeypatch.setenv("LZ_ROOT_PUCT_DIV", "1")
    v_d, vs_d, rv_d = (t.cpu().numpy() for t in v0.root_puct_allocate_visits(*args, sims, 1.5))
    assert np.array_equal(v_t, v_d)
    assert vs_t.tobytes() == vs_d.tobytes() and rv_t.tobytes() == rv_d.tobytes()
    live = np.isfinite(v_t).all(1)
    assert (v_t[live].sum(1) <= sims).all() and (v_t[0].sum() == sims)
    # rows packed to the left, as the fused root search builds them: two roots share a wave when both have <= 32 actions
    # (pairs of small rows, mixed pairs, an odd last row, rows with no action at all)
    R2 = 95
    n = rng.integers(0, min(A, 44) + 1, R2)
    n[:8] = [1, 32, 32, 33, 0, 5, 31, 0]
    valid2 = np.arange(A)[None, :] < n[:, None]
    pri2 = (rng.random((R2, A)) ** 2 * valid2).astype(np.float32)
    pri2 /= np.maximum(pri2.sum(1, keepdims=True), 1e-8)
    leaf2 = ((rng.random((R2, A)) * 2 - 1) * valid2).astype(np.float32)
    leaf2[10, :2] = [np.float32(2.0 ** -120), 0.5]                          # a pair that must take the division branch
    args2 = [torch.from_numpy(x).to(DEV) for x in (pri2, leaf2, valid2)]
    s2 = min(sims, 4096)
    monkeypatch.delenv("LZ_ROOT_PUCT_DIV", raising=False)
    a_t = [t.cpu().numpy() for t in v0.root_puct_allocate_visits(*args2, s2, 1.25)]
    monkeypatch.setenv("LZ_ROOT_PUCT_DIV", "1")
    a_d = [t.cpu().numpy() for t in v0.root_puct_allocate_visits(*args2, s2, 1.25)]
    for x, y in zip(a_t, a_d):
        assert x.tobytes() == y.tobytes()
    assert (a_t[0].sum(1) == np.where(n > 0, s2, 0)).all() and (a_t[0][~valid2] == 0).all()
    want_v, _, _ = O.root_puct(pri2, leaf2, valid2, min(s2, 300), 1.25)     # and against the oracle at a budget it can follow
    monkeypatch.delenv("LZ_ROOT_PUCT_DIV", raising=False)
    got_v = v0.root_puct_allocate_visits(*args2, min(s2, 300), 1.25)[0].cpu().numpy()
    assert np.array_equal(got_v, want_v)


@pytest.mark.parametrize("R,A,sims", [(4, 36, 64), (5, 72, 1024), (1023, 80, 700), (6, 16, 8192), (130, 130, 300)])
def test_root_puct_binned_by_width_equals_neighbour_pairs_and_the_division_kernel(v0, R, A, sims, monkeypatch):
    """Round 5: rows of <= 8 valid actions go EIGHT to a wave, <= 16 four, <= 32 two, the rest alone, whichever roots they
    are (csrc/lz_ops.hip: puct_bin_kernel + root_puct_binned_kernel).  Byte-identical visits / value sums / root values to
    the round-4 kernel (LZ_ROOT_PUCT_BIN=0: neighbours pair up) and to the IEEE-division kernel (LZ_ROOT_PUCT_DIV=1), over
    widths 0 / 1 / 7 / 8 / 9 / 15 / 16 / 17 / 31 / 32 / 33 / 64+, root counts that leave incomplete octets, quadruples and pairs, ties,
    rows that must take the division branch, NaN scores, and the same call captured into a hipGraph and replayed."""
    if DEV == "cpu":
        pytest.skip("the host build has one loop per root: nothing to bin")
    rng = np.random.default_rng(R * 1000 + A)
    special = [0, 1, 15, 16, 17, 31, 32, 33, min(A, 64), A, 16, 16, 16, 3, 16, 8, 9, 7, 8, 8, 2, 8, 8, 8, 8, 5]
    n = np.minimum(np.where(np.arange(R) < len(special), np.resize(special, R), rng.integers(0, min(A, 40) + 1, R)), A)
    narrow = rng.random(R) < 0.5
    n[narrow & (np.arange(R) >= len(special))] = np.minimum(n, 16)[narrow & (np.arange(R) >= len(special))]     # most rows are narrow
    tiny = rng.random(R) < 0.3
    n[tiny & (np.arange(R) >= len(special))] = np.minimum(n, 8)[tiny & (np.arange(R) >= len(special))]           # ... a third of them <= 8
    valid = np.arange(A)[None, :] < n[:, None]
    holes = rng.random((R, A)) < 0.15                               # not every row is packed to the left
    valid = valid & ~holes
    valid[np.arange(R), np.maximum(n - 1, 0)] = n > 0               # ... but its width is what it is
    pri = (rng.random((R, A)) ** 2 * valid).astype(np.float32)
    pri /= np.maximum(pri.sum(1, keepdims=True), 1e-8)
    leaf = ((rng.random((R, A)) * 2 - 1) * valid).astype(np.float32)
    if R > 12:
        pri[11, :] = np.where(valid[11], 1.0 / max(1, valid[11].sum()), 0).astype(np.float32); leaf[11] = 0.25 * valid[11]   # ties
        leaf[12, 0] = np.float32(2.0 ** -120); valid[12, 0] = True                                   # division branch
    dead = np.zeros(R, dtype=bool)
    if R > 15:                                                      # scores that are NaN from the start / become NaN / a root
        pri[13, 0] = np.nan; valid[13, 0] = True                    # without any usable score inside a packed wave (the pull
        leaf[14, 1] = np.nan; valid[14, 1] = True                   # loops have no early exit: nothing may be pulled there)
        pri[10] = np.nan; dead[10] = True
    args = [torch.from_numpy(x).to(DEV) for x in (pri, leaf, valid)]
    run = lambda: [t.cpu().numpy() for t in v0.root_puct_allocate_visits(*args, sims, 1.3)]
    monkeypatch.delenv("LZ_ROOT_PUCT_DIV", raising=False)
    monkeypatch.delenv("LZ_ROOT_PUCT_BIN", raising=False)
    binned = run()
    monkeypatch.setenv("LZ_ROOT_PUCT_BIN", "0")
    pairs = run()
    monkeypatch.setenv("LZ_ROOT_PUCT_DIV", "1")
    division = run()
    for x, y, z in zip(binned, pairs, division):
        assert x.tobytes() == y.tobytes() == z.tobytes()
    width = np.where(valid.any(1), A - np.argmax(valid[:, ::-1], axis=1), 0)
    assert (binned[0].sum(1) == np.where((width > 0) & ~dead, sims, 0)).all() and (binned[0][~valid] == 0).all()
    # the same call inside a captured graph (the scratch lists exist since the eager call above)
    monkeypatch.delenv("LZ_ROOT_PUCT_DIV", raising=False)
    monkeypatch.delenv("LZ_ROOT_PUCT_BIN", raising=False)
    from liuzhou_amd import _lib as L
    import ctypes as C
    vis, vs, rv = (torch.zeros((R, A), device=DEV), torch.zeros((R, A), device=DEV), torch.zeros((R,), device=DEV))
    nbytes = C.c_int64(0)
    L.check(L.lib().lz_root_puct_workspace_bytes(L.i64(R), C.byref(nbytes)), "workspace_bytes")
    ws = torch.zeros((int(nbytes.value),), dtype=torch.uint8, device=DEV)      # caller-owned lists: allocation-free launch
    call = lambda: L.check(L.lib().lz_root_puct_allocate_visits_ws(
        L.ptr(args[0]), L.ptr(args[1]), L.ptr(args[2]), L.i64(R), L.i64(A), L.i64(sims), C.c_float(1.3), L.ptr(vis),
        L.ptr(vs), L.ptr(rv), L.ptr(ws), L.i64(int(ws.numel())), L.stream_ptr(torch.device(DEV))), "root_puct_ws")
    call()
    torch.cuda.synchronize()
    for x, y in zip(binned, (vis, vs, rv)):
        assert x.tobytes() == y.cpu().numpy().tobytes()
    vis.zero_(); vs.zero_(); rv.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                                       # torch captures on a stream of its own
        call()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    small = torch.zeros((8,), dtype=torch.uint8, device=DEV)
    assert L.lib().lz_root_puct_allocate_visits_ws(
        L.ptr(args[0]), L.ptr(args[1]), L.ptr(args[2]), L.i64(R), L.i64(A), L.i64(sims), C.c_float(1.3), L.ptr(vis),
        L.ptr(vs), L.ptr(rv), L.ptr(small), L.i64(8), L.stream_ptr(torch.device(DEV))) == -1     # LZ_ERR_ARG
    for x, y in zip(binned, (vis, vs, rv)):
        assert x.tobytes() == y.cpu().numpy().tobytes()


@pytest.mark.parametrize("A", [5, 64, 72, 130])
def test_root_puct_random_vs_oracle(v0, A):
    rng = np.random.default_rng(A)
    R = 512
    valid = rng.random((R, A)) < 0.5
    valid[:, rng.integers(0, A)] = True
    pri = (rng.random((R, A)) * valid).astype(np.float32)
    pri /= np.maximum(pri.sum(1, keepdims=True), 1e-8)
    leaf = ((rng.random((R, A)) * 2 - 1) * valid).astype(np.float32)
    want_v, want_vs, want_rv = O.root_puct(pri, leaf, valid, 200, 1.25)
    v, vs, rv = v0.root_puct_allocate_visits(torch.from_numpy(pri).to(DEV), torch.from_numpy(leaf).to(DEV),
                                             torch.from_numpy(valid).to(DEV), 200, 1.25)
    assert np.array_equal(v.cpu().numpy(), want_v)
    np.testing.assert_allclose(vs.cpu().numpy(), want_vs, atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), want_rv, atol=1e-5, rtol=0)


def test_root_finalize_from_visits(v0):
    z = load("g7_ops.npz")
    g = lambda k: torch.from_numpy(z[k]).to(DEV)
    B = z["s_board"].shape[0]
    out = v0.root_finalize_from_visits(g("pack_legal_index_mat"), g("pack_action_code_mat"), g("pack_valid_mask"),
                                       g("fin_visits"), g("fin_value_sum"), g("pack_valid_root_indices"), B, 220,
                                       g("fin_temps"), False)
    np.testing.assert_allclose(out[0].cpu().numpy(), z["fin_policy_dense"], atol=1e-5, rtol=0)
    assert np.array_equal(out[1].cpu().numpy(), z["fin_chosen_idx"])
    assert np.array_equal(out[2].cpu().numpy(), z["fin_chosen_codes"])
    assert np.array_equal(out[3].cpu().numpy(), z["fin_chosen_valid"])
    np.testing.assert_allclose(out[4].cpu().numpy(), z["fin_root_value"], atol=1e-6, rtol=0)
    # sampled picks with injected uniforms: must be legal, and follow the inverse CDF of the stable policy
    R, M = z["fin_visits"].shape
    u = torch.linspace(0.01, 0.99, R).to(DEV)
    out_s = v0.root_finalize_from_visits(g("pack_legal_index_mat"), g("pack_action_code_mat"), g("pack_valid_mask"),
                                         g("fin_visits"), g("fin_value_sum"), g("pack_valid_root_indices"), B, 220,
                                         g("fin_temps"), True, uniforms=u)
    picks = out_s[1].cpu().numpy()
    vis = z["fin_visits"]; valid = z["pack_valid_mask"]; temps = z["fin_temps"]
    for r in range(R):
        b = int(z["pack_valid_root_indices"][r])
        logits = np.where(valid[r], np.log(np.maximum(vis[r], 1e-8)) / max(temps[r], 1e-6), -np.inf)
        e = np.where(valid[r], np.exp(logits - logits.max()), 0.0)
        cdf = np.cumsum(e)
        target = float(u[r].item()) * cdf[-1]
        k = int(np.searchsorted(cdf, target, side="right"))
        k = min(k, int(np.nonzero(e > 0)[0].max()))
        cand = {int(z["pack_legal_index_mat"][r, k])}
        if k > 0 and abs(cdf[k - 1] - target) < 1e-5 * cdf[-1]:
            cand.add(int(z["pack_legal_index_mat"][r, k - 1]))
        if k + 1 < M and abs(cdf[k] - target) < 1e-5 * cdf[-1]:
            cand.add(int(z["pack_legal_index_mat"][r, k + 1]))
        assert int(picks[b]) in cand, (r, picks[b], cand)
    np.testing.assert_allclose(out_s[0].cpu().numpy(), z["fin_policy_dense"], atol=1e-5, rtol=0)


def test_self_play_step_inplace(v0):
    z = load("g7_ops.npz")
    t = to_dev(states(z, "s"))
    plies = torch.from_numpy(z["step_plies_in"].copy()).to(DEV)
    done = torch.zeros(plies.shape[0], dtype=torch.bool, device=DEV)
    g = lambda k: torch.from_numpy(z[k]).to(DEV)
    slots, res, soft = v0.self_play_step_inplace(*t, plies, done, g("step_active"), g("step_codes"), g("step_term"),
                                                 g("step_valid"), 96, 2.0)
    assert np.array_equal(slots.cpu().numpy(), z["step_slots"])
    assert np.array_equal(res.cpu().numpy(), z["step_result"])
    np.testing.assert_allclose(soft.cpu().numpy(), z["step_soft"], atol=1e-6, rtol=0)
    assert np.array_equal(plies.cpu().numpy(), z["step_plies_out"])
    assert np.array_equal(done.cpu().numpy(), z["step_done_out"])
    ok, field = states_equal(from_dev(t), {f: z[f"step_after_{f}"] for f in FIELDS})
    assert ok, field


def test_finalize_trajectory_inplace(v0):
    z = load("g7_ops.npz")
    g = lambda k: torch.from_numpy(z[k]).to(DEV)
    S = z["traj_value_out"].shape[0]
    vt = torch.full((S,), float("nan"), device=DEV); svt = torch.full((S,), float("nan"), device=DEV)
    fs, fc, co = v0.finalize_trajectory_inplace(vt, svt, g("traj_signs"), g("traj_step_index"), g("traj_counts"),
                                                g("traj_slots"), g("traj_result"), g("traj_soft"))
    np.testing.assert_array_equal(vt.cpu().numpy(), z["traj_value_out"])
    np.testing.assert_allclose(svt.cpu().numpy(), z["traj_soft_out"], atol=1e-7, rtol=0, equal_nan=True)
    assert np.array_equal(fs.cpu().numpy(), z["traj_final_slots"])
    assert np.array_equal(fc.cpu().numpy(), z["traj_final_counts"])
    assert np.array_equal(co.cpu().numpy(), z["traj_counts_out"])


def test_cpu_and_hip_tensors_dispatch_to_their_own_build(v0):
    """Device dispatch like the reference extension (fast_legal_mask.cpp:453): the same call on CPU tensors runs the host
    build and returns CPU tensors with the same contents as the HIP kernels."""
    z = load("g1_rules.npz")
    st = states(z, "s")
    t_dev = to_dev(st)
    t_cpu = [t.cpu() for t in t_dev]
    m1, d1 = v0.encode_actions_fast(*t_dev[:10], 36, 144, 36, 4)
    m2, d2 = v0.encode_actions_fast(*t_cpu[:10], 36, 144, 36, 4)
    assert m1.is_cuda and not m2.is_cuda
    assert torch.equal(m1.cpu(), m2) and torch.equal(d1.cpu(), d2)


def test_model_fp32_matches_reference_outputs(v0):
    """<= 1e-5 on policy / value tensors (fp32 path), weights regenerated from the seed on the box."""
    import json, os
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from tests.golden_utils import GOLDEN
    z = load("g9_net.npz")
    keys = json.load(open(os.path.join(GOLDEN, "g9_net_keys.json")))
    x = torch.from_numpy(z["inputs"].astype(np.float32)).to(DEV)
    for name, seed in (("tiny", 7), ("b6c64", 20260314), ("b10c128", 20260314)):
        torch.manual_seed(seed)
        m = ChessNet(**MODEL_CONFIGS[name])
        gen = torch.Generator().manual_seed(seed + 1)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) * 0.5 + 0.75)
                mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=gen) * 0.5 + 0.75)
                mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.1)
        assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, s) for k, s in keys[name]]
        m = m.to(DEV).eval()
        with torch.inference_mode():
            out = m(x)
        for got, k in zip(out, ("lp1", "lp2", "lpmc", "value_logits")):
            np.testing.assert_allclose(got.cpu().numpy(), z[f"{name}_{k}"], atol=1e-5, rtol=0, err_msg=f"{name}/{k}")


def test_root_force_uniform_picks():
    """lz_root_force_uniform_picks (opening plies, mcts_gpu.py:1425-1447): the k-th valid slot, k = min(floor(u n), n - 1),
    only for flagged roots; unflagged roots and rows without a valid slot keep the search's pick."""
    if not torch.cuda.is_available() or DEV == "cpu":
        pytest.skip("HIP only")
    from liuzhou_amd import _lib as L
    rng = np.random.default_rng(9)
    B, R, M = 300, 260, 80
    roots = np.sort(rng.choice(B, R, replace=False)).astype(np.int64)
    valid = rng.random((R, M)) < 0.3
    valid[0] = False; valid[1] = True; valid[2] = False; valid[2, 79] = True
    lidx = rng.integers(0, 220, (R, M)).astype(np.int64)
    codes = rng.integers(-1, 36, (R, M, 4)).astype(np.int32)
    force = rng.random(B) < 0.6
    force[roots[:3]] = True
    u = rng.random(R).astype(np.float32); u[1] = 0.999999; u[3] = 0.0
    cidx0 = rng.integers(0, 220, B).astype(np.int64); cc0 = rng.integers(0, 9, (B, 4)).astype(np.int32)
    cv0 = np.zeros(B, bool)
    t = lambda x: torch.from_numpy(x).to(DEV)
    d = dict(lidx=t(lidx), codes=t(codes), valid=t(valid), roots=t(roots), force=t(force), u=t(u), cidx=t(cidx0.copy()),
             cc=t(cc0.copy()), cv=t(cv0.copy()))
    L.check(L.lib().lz_root_force_uniform_picks(L.ptr(d["lidx"]), L.ptr(d["codes"]), L.ptr(d["valid"]), L.ptr(d["roots"]),
                                                L.i64(R), L.i64(M), L.ptr(d["force"]), L.ptr(d["u"]), L.ptr(d["cidx"]),
                                                L.ptr(d["cc"]), L.ptr(d["cv"]), L.stream_ptr(torch.device(DEV))), "force")
    want_idx, want_cc, want_cv = cidx0.copy(), cc0.copy(), cv0.copy()
    for r in range(R):
        b = roots[r]
        slots = np.nonzero(valid[r])[0]
        if not force[b] or len(slots) == 0:
            continue
        k = min(int(np.float32(u[r]) * np.float32(len(slots))), len(slots) - 1)
        want_idx[b], want_cc[b], want_cv[b] = lidx[r, slots[k]], codes[r, slots[k]], True
    assert np.array_equal(d["cidx"].cpu().numpy(), want_idx) and np.array_equal(d["cc"].cpu().numpy(), want_cc)
    assert np.array_equal(d["cv"].cpu().numpy(), want_cv)
