"""CPU: pin the oracle (oracle/lz_oracle.c + numpy host ops) against the reference's golden vectors."""
import numpy as np
import pytest

from oracle import lz_oracle as O
from tests.golden_utils import load, states, unpack_mask, states_equal, FIELDS, row_hash64, group_hash64, state_rows


def test_g1_legal_masks_and_metadata_bit_exact():
    z = load("g1_rules.npz")
    st = states(z, "s")
    mask, meta = O.encode_actions(st)
    assert np.array_equal(mask, unpack_mask(z["legal_mask"], 220))
    assert np.array_equal(meta, z["metadata"].astype(np.int32))


def test_g1_every_child_transition_bit_exact():
    z = load("g1_rules.npz")
    st = states(z, "s")
    _, meta = O.encode_actions(st)
    parents = z["child_parent"].astype(np.int64)
    actions = z["child_action"].astype(np.int64)
    codes = meta[parents, actions]
    out = O.apply_moves(st, codes, parents, strict=True)
    ok, field = states_equal(out, states(z, "c"))
    assert ok, field
    # GPU (no-op) semantics agree on legal actions
    out2 = O.apply_moves(st, codes, parents, strict=False)
    ok, field = states_equal(out2, out)
    assert ok, field


def test_g1_python_semantics_index_api():
    z = load("g1_rules.npz")
    st = states(z, "s")
    want = unpack_mask(z["legal_mask"], 220)
    for i in range(0, st["board"].shape[0], 7):
        cs = O.state_from_batch(st, i)
        assert O.legal_indices_py(cs) == list(np.nonzero(want[i])[0])


def test_g2_terminal_and_representative_states():
    z = load("g2_edges.npz")
    st = states(z, "s")
    py_mask = unpack_mask(z["py_legal_mask"], 220)
    for i in range(st["board"].shape[0]):
        cs = O.state_from_batch(st, i)
        assert O.game_status(cs) == int(z["status"][i])
        got = np.zeros(220, bool); got[O.legal_indices_py(cs)] = True
        assert np.array_equal(got, py_mask[i]), i
    mask, meta = O.encode_actions(st)
    assert np.array_equal(mask, unpack_mask(z["tensor_mask"], 220))
    assert np.array_equal(meta, z["tensor_meta"].astype(np.int32))
    assert np.array_equal(O.states_to_model_input(st), z["model_input"].astype(np.float32))


@pytest.mark.parametrize("aux,key", [(1, "t217"), (4, "t220")])
def test_g3_garbage_states_match_reference_op(aux, key):
    z = load("g3_garbage.npz")
    st = states(z, "s")
    mask, meta = O.encode_actions(st, 36, 144, 36, aux)
    T = 216 + aux
    assert np.array_equal(mask, unpack_mask(z[f"mask_{key}"], T))
    assert np.array_equal(meta, z[f"meta_{key}"].astype(np.int32))
    assert np.array_equal(O.states_to_model_input(st), z["model_input"].astype(np.float32))


# ---- reference-scale suites (oracle/gen_golden_large.py) ----
@pytest.mark.parametrize("aux,key", [(1, "t217"), (4, "t220")])
def test_g16_reference_mask_suite_10000_states(aux, key):
    z = load("g16_garbage_large.npz")
    st = states(z, "s")
    mask, meta = O.encode_actions(st, 36, 144, 36, aux)
    assert mask.shape[0] == 10000
    assert np.array_equal(mask, unpack_mask(z[f"mask_{key}"], 216 + aux))
    assert np.array_equal(row_hash64(meta), z[f"meta_hash_{key}"])
    if aux == 4:
        assert np.array_equal(row_hash64(O.states_to_model_input(st).astype(np.int8)), z["model_input_hash"])


def test_g17_reference_apply_suite_10000_micro_positions():
    z = load("g17_apply_micro.npz")
    st = states(z, "s")
    n = st["board"].shape[0]
    codes = z["codes"].astype(np.int32)
    out = O.apply_moves(st, codes, np.arange(n, dtype=np.int64), strict=True)       # every action is legal: nothing raises
    ok, field = states_equal(out, states(z, "c"))
    assert ok, field
    ok, field = states_equal(O.apply_moves(st, codes, np.arange(n, dtype=np.int64), strict=False), states(z, "c"))
    assert ok, field


def test_g15_reference_playout_suite_5000_states():
    z = load("g15_rules_large.npz")
    st = states(z, "s")
    n = st["board"].shape[0]
    assert n >= 5000
    mask, meta = O.encode_actions(st)
    want = unpack_mask(z["legal_mask"], 220)
    assert np.array_equal(mask, want)
    assert np.array_equal(row_hash64(meta), z["metadata_hash"])
    parents, actions = np.nonzero(want)
    assert parents.size == int(z["num_children"])
    out = O.apply_moves(st, meta[parents, actions], parents.astype(np.int64), strict=True)
    assert np.array_equal(group_hash64(row_hash64(state_rows(out)), parents, n), z["children_hash"])
    for i in range(0, n, 97):                                   # Python-semantics index API on a sample
        assert O.legal_indices_py(O.state_from_batch(st, i)) == list(np.nonzero(want[i])[0])


def test_row_hash_helpers_detect_a_single_changed_element():
    rng = np.random.default_rng(0)
    a = rng.integers(-5, 40, (64, 220, 4))
    h = row_hash64(a)
    b = a.copy(); b[17, 200, 3] += 1
    assert np.flatnonzero(row_hash64(b) != h).tolist() == [17]
    items = rng.integers(0, 1 << 62, 30).astype(np.uint64)
    grp = np.sort(rng.integers(0, 8, 30))
    g = group_hash64(items, grp, 8)
    sw = items.copy()
    j = int(np.flatnonzero(np.diff(grp) == 0)[0])              # swap two neighbours of one group: order matters
    sw[[j, j + 1]] = sw[[j + 1, j]]
    assert np.flatnonzero(group_hash64(sw, grp, 8) != g).tolist() == [int(grp[j])]


def test_g4_policy_projection():
    z = load("g4_project.npz")
    mask = unpack_mask(z["mask"], 220)
    probs, ml = O.project_policy(z["lp1"], z["lp2"], z["lpmc"], mask)
    np.testing.assert_allclose(probs, z["probs"], atol=1e-6, rtol=0)
    want = z["masked_logits"]
    assert np.array_equal(np.isneginf(ml), np.isneginf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(ml[fin], want[fin], atol=1e-6, rtol=0)


def _run_oracle_case(z, ci, est):
    root_states = states(z, "r")
    ri = int(z["case_root"][ci]); sims = int(z["case_sims"][ci])
    start = int(z["case_eval_start"][ci]); count = int(z["case_eval_count"][ci])
    noise = z["case_noise"][ci] if bool(z["case_noise_flag"][ci]) else None
    tree = O.OracleTree(O.state_from_batch(root_states, ri), 1.0)
    k = start

    def check_pending():
        pend = O.batch_from_states([tree.pending_state()])
        for f in FIELDS:
            assert np.array_equal(np.asarray(pend[f]).reshape(-1).astype(np.int64),
                                  np.asarray(est[f][k]).reshape(-1).astype(np.int64)), (ci, k, f)

    if tree.prepare_root():
        check_pending()
        tree.complete(z["eval_priors"][k], float(z["eval_value"][k]), noise, 0.25)
        k += 1
    for _ in range(sims):
        if tree.select():
            check_pending()
            tree.complete(z["eval_priors"][k], float(z["eval_value"][k]))
            k += 1
    assert k == start + count, "oracle requested a different number of evaluations"
    return tree


def test_g5_tree_visit_counts_bit_exact_vs_portable_and_legacy():
    z = load("g5_tree.npz")
    est = states(z, "e")
    n_cases = z["case_root"].shape[0]
    for ci in range(n_cases):
        tree = _run_oracle_case(z, ci, est)
        idx, vis, vs, pr, pl = tree.root_children()
        got = np.zeros(220, np.int32); got[idx] = vis
        assert np.array_equal(got, z["case_visits"][ci]), ci
        assert int(vis.sum()) == int(z["case_sims"][ci])
        pri = np.zeros(220, np.float32); pri[idx] = pr
        np.testing.assert_allclose(pri, z["case_root_priors"][ci], atol=1e-6, rtol=0)
        p1 = np.zeros(220, np.float32); p1[idx] = O.policy_from_visits(vis, 1.0)
        np.testing.assert_allclose(p1, z["case_policy_t1"][ci], atol=1e-6, rtol=0)
        p01 = np.zeros(220, np.float32); p01[idx] = O.policy_from_visits(vis, 0.1)
        np.testing.assert_allclose(p01, z["case_policy_t01"][ci], atol=1e-6, rtol=0)
        rv = tree.root_value_sum() / max(1, tree.root_visits())
        assert abs(rv - float(z["case_root_value"][ci])) < 1e-6
    # src/mcts.py (batch_K=1) agrees with the same counts
    for j, ci in enumerate(z["legacy_case"]):
        assert np.array_equal(z["legacy_visits"][j], z["case_visits"][int(ci)])


def test_g6_root_puct_bit_exact():
    z = load("g6_root_puct.npz")
    for sims in (1, 16, 200, 1024):
        v, vs, rv = O.root_puct(z["priors"], z["leaf"], z["valid"], sims, 1.0)
        assert np.array_equal(v, z[f"visits_{sims}"]), sims
        np.testing.assert_allclose(vs, z[f"value_sum_{sims}"], atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(rv, z[f"root_{sims}"], atol=1e-5, rtol=0)
    v, vs, _ = O.root_puct(z["priors"], z["leaf"], z["valid"], 64, 2.5)
    assert np.array_equal(v, z["visits_64_c25"])


def test_g7_host_ops():
    z = load("g7_ops.npz")
    st = states(z, "s")
    mask = unpack_mask(z["mask"], 220)
    meta = z["meta"].astype(np.int32)
    pack = O.root_pack_sparse_actions(mask, z["probs"], meta)
    names = ["terminal_mask", "valid_root_indices", "counts", "valid_mask", "legal_index_mat", "priors_mat",
             "action_code_mat", "pack_flat_idx", "action_codes_all", "parent_indices_all"]
    for n, got in zip(names, pack):
        want = z[f"pack_{n}"]
        if got.dtype == np.float32:
            np.testing.assert_allclose(got, want, atol=1e-6, rtol=0, err_msg=n)
        else:
            assert np.array_equal(got, want), n
    fin = O.root_finalize_from_visits(pack[4], pack[6], pack[3], z["fin_visits"], z["fin_value_sum"], pack[1],
                                      st["board"].shape[0], 220, z["fin_temps"])
    np.testing.assert_allclose(fin[0], z["fin_policy_dense"], atol=1e-5, rtol=0)
    assert np.array_equal(fin[1], z["fin_chosen_idx"])
    assert np.array_equal(fin[2], z["fin_chosen_codes"])
    assert np.array_equal(fin[3], z["fin_chosen_valid"])
    np.testing.assert_allclose(fin[4], z["fin_root_value"], atol=1e-6, rtol=0)

    work = {f: np.array(st[f]) for f in FIELDS}
    plies = z["step_plies_in"].copy(); done = np.zeros(plies.shape[0], bool)
    slots, res, soft = O.self_play_step_inplace(work, plies, done, z["step_active"], z["step_codes"],
                                                z["step_term"], z["step_valid"], 96, 2.0)
    assert np.array_equal(slots, z["step_slots"])
    assert np.array_equal(res, z["step_result"])
    np.testing.assert_allclose(soft, z["step_soft"], atol=1e-6, rtol=0)
    assert np.array_equal(plies, z["step_plies_out"]) and np.array_equal(done, z["step_done_out"])
    after = {f: z[f"step_after_{f}"] for f in FIELDS}
    ok, field = states_equal(work, after)
    assert ok, field

    vt = np.full(z["traj_value_out"].shape, np.nan, np.float32); svt = vt.copy()
    fs, fc, co = O.finalize_trajectory_inplace(vt, svt, z["traj_signs"], z["traj_step_index"], z["traj_counts"],
                                               z["traj_slots"], z["traj_result"], z["traj_soft"])
    np.testing.assert_array_equal(vt, z["traj_value_out"])
    np.testing.assert_allclose(svt, z["traj_soft_out"], atol=1e-7, rtol=0, equal_nan=True)
    assert np.array_equal(fs, z["traj_final_slots"]) and np.array_equal(fc, z["traj_final_counts"])
    assert np.array_equal(co, z["traj_counts_out"])
