"""CPU ORACLE of the self-play wave loop (test infrastructure / bench cpu_baseline only).

Restates, on top of oracle/lz_oracle.c + numpy host ops + a PyTorch-CPU network:
  - variant R: v1/python/mcts_gpu.py:1249-1457 + v1/python/self_play_gpu_runner.py:159-256
  - variant P: v1/python/portable_mcts.py:570-700 + v1/python/portable_self_play.py:82-284
Pinned by tests/golden/g8_selfplay.npz (the reference's own v1 runner on CPU, 4 games x 32 sims).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import lz_oracle as O


def _net_eval(model, planes: np.ndarray):
    with torch.inference_mode():
        lp1, lp2, lpm, raw = model(torch.from_numpy(planes))
        probs = torch.softmax(raw.float(), dim=-1)
        centers = torch.linspace(-1.0, 1.0, steps=raw.shape[-1], dtype=probs.dtype)
        val = (probs * centers).sum(-1)
    return lp1.float().numpy(), lp2.float().numpy(), lpm.float().numpy(), val.numpy().astype(np.float32)


def _children_leaf(model, states: Dict[str, np.ndarray], soft_k: float):
    """Every legal child of every state evaluated, as seen by the parent's mover (mcts_gpu.py:900-975,
    :1341-1375): dict with the packed-root tensors, leaf_mat [R,M] and the two eval counts."""
    B = states["board"].shape[0]
    planes = O.states_to_model_input(states)
    lp1, lp2, lpm, values = _net_eval(model, planes)
    mask, meta = O.encode_actions(states)
    probs, _ = O.project_policy(lp1, lp2, lpm, mask)
    (term, roots, counts, valid, lidx, pri, codes, flat, codes_all, parents_all) = O.root_pack_sparse_actions(mask, probs, meta)
    out = dict(planes=planes, mask=mask, values=values, term=term, roots=roots, counts=counts, valid=valid, lidx=lidx,
               pri=pri, codes=codes, evals=B, leaf=None)
    if roots.size == 0:
        return out
    R, M = valid.shape
    child = O.apply_moves(states, codes_all, parents_all, strict=True)
    _, _, _, cvals = _net_eval(model, O.states_to_model_input(child))
    out["evals"] += int(cvals.shape[0])
    parent_player = states["current_player"][parents_all]
    leaf = np.where(child["current_player"] == parent_player, cvals, -cvals).astype(np.float32)
    tchild = O.terminal_mask_from_next_state(child)
    soft = O.soft_value_from_board(child["board"], soft_k)
    sign = np.where(parent_player >= 0, 1.0, -1.0).astype(np.float32)
    leaf = np.where(tchild, soft * sign, leaf).astype(np.float32)
    leaf_mat = np.zeros((R, M), np.float32)
    leaf_mat.reshape(-1)[flat] = leaf
    out["leaf"] = leaf_mat
    return out


def _refine_topk(model, root_states: Dict[str, np.ndarray], leaf_mat: np.ndarray, valid: np.ndarray,
                 codes: np.ndarray, top_k: int, soft_k: float):
    """One more ply below the K best children of every root (mcts_gpu.py:976-1046): the child's value becomes
    max(its value, best value among ITS children as its own mover sees them).  Returns (leaf_mat, evals)."""
    R, M = valid.shape
    K = min(int(top_k), M)
    if K <= 0 or R <= 0:
        return leaf_mat, 0
    masked = np.where(valid, leaf_mat, -np.inf).astype(np.float32)
    top = np.argsort(-masked, axis=1, kind="stable")[:, :K]                       # [R,K] local action slots
    picked_valid = np.take_along_axis(valid, top, axis=1)                          # fewer than K legal actions
    # slots past the legal count carry no action: evaluate the root's first action there and discard the result
    safe = np.where(picked_valid, top, top[:, :1])
    l2_codes = np.take_along_axis(codes, safe[:, :, None], axis=1).reshape(-1, 4)
    parents = np.repeat(np.arange(R, dtype=np.int64), K)
    l2 = O.apply_moves(root_states, l2_codes, parents, strict=True)
    sub = _children_leaf(model, l2, soft_k)
    refined = np.zeros(R * K, np.float32)                                          # no grandchildren -> 0
    if sub["leaf"] is not None:
        best = np.where(sub["valid"], sub["leaf"], -np.inf).max(axis=1)
        refined[sub["roots"]] = np.where(np.isfinite(best), best, 0.0).astype(np.float32)
    refined = refined.reshape(R, K)
    original = np.take_along_axis(leaf_mat, top, axis=1)
    new = np.maximum(original, refined).astype(np.float32)
    out = leaf_mat.copy()
    rr, kk = np.nonzero(picked_valid)
    out[rr, top[rr, kk]] = new[rr, kk]
    return out, sub["evals"]


def root_search_batch(model, states: Dict[str, np.ndarray], temps: np.ndarray, sims: int, c: float,
                      soft_k: float = 2.0, noise: Optional[np.ndarray] = None, eps: float = 0.25,
                      sparse_ply: int = 1, sparse_top_k: int = 8):
    """variant R on a batch (sample_moves=False): returns dict with model_input, legal_mask, policy_dense,
    chosen codes/idx/valid, terminal_mask, visits matrices.  `sparse_ply` > 1: the reference's experimental top-K
    lookahead refines the children's values before the bandit (mcts_gpu.py:1048-1160)."""
    B = states["board"].shape[0]
    ch = _children_leaf(model, states, soft_k)
    planes, mask, values, term, roots = ch["planes"], ch["mask"], ch["values"], ch["term"], ch["roots"]
    out = dict(model_input=planes, legal_mask=mask, policy_dense=np.zeros((B, 220), np.float32),
               chosen_idx=np.full(B, -1, np.int64), chosen_codes=np.full((B, 4), -1, np.int32),
               chosen_valid=np.zeros(B, bool), terminal_mask=term, root_value=values.copy(), leaf_evals=ch["evals"])
    if roots.size == 0:
        return out
    counts, valid, lidx, pri, codes, leaf_mat = ch["counts"], ch["valid"], ch["lidx"], ch["pri"], ch["codes"], ch["leaf"]
    R, M = valid.shape
    if noise is not None and M > 1:
        nz = noise[:R, :M].astype(np.float32) * valid
        nz = nz / np.maximum(nz.sum(1, keepdims=True), np.float32(1e-8))
        mixed = (np.float32(1.0 - eps) * pri + np.float32(eps) * nz).astype(np.float32)
        pri = np.where((counts > 1)[:, None], mixed, pri)
    if int(sparse_ply) > 1:
        root_states = O.select_states(states, roots)
        for _ in range(2, int(sparse_ply) + 1):
            leaf_mat, more = _refine_topk(model, root_states, leaf_mat, valid, codes, sparse_top_k, soft_k)
            out["leaf_evals"] += more
    visits, vsum, _ = O.root_puct(pri, leaf_mat, valid, sims, c)
    pol, cidx, ccodes, cvalid, rv = O.root_finalize_from_visits(lidx, codes, valid, visits, vsum, roots, B, 220,
                                                                temps[roots])
    out.update(policy_dense=pol, chosen_idx=cidx, chosen_codes=ccodes, chosen_valid=cvalid, visits=visits,
               valid=valid, lidx=lidx, roots=roots, priors=pri, leaf=leaf_mat)
    out["root_value"][roots] = rv
    return out


def self_play_root(model, num_games: int, sims: int, temperature_init: float = 1.0, temperature_final: float = 0.1,
                   temperature_threshold: int = 10, c: float = 1.0, soft_k: float = 2.0, max_game_plies: int = 512,
                   max_total_plies: Optional[int] = None, time_budget_s: Optional[float] = None,
                   sparse_ply: int = 1, sparse_top_k: int = 8):
    """variant-R wave loop, deterministic (sample_moves=False, no noise).  Returns (tensors dict, stats dict)."""
    states = O.initial_states(num_games)
    plies = np.zeros(num_games, np.int64); done = np.zeros(num_games, bool)
    step_index = np.full((num_games, max_game_plies), -1, np.int64); step_counts = np.zeros(num_games, np.int64)
    S, L, P, signs = [], [], [], []
    size = 0
    finished = []
    leaf_evals = 0
    t0 = time.perf_counter()
    waves = 0
    while True:
        active = np.nonzero(~done)[0]
        if active.size == 0 or (max_total_plies is not None and waves >= max_total_plies):
            break
        if time_budget_s is not None and waves > 0 and time.perf_counter() - t0 > time_budget_s:
            break
        act = O.select_states(states, active)
        temps = np.where(plies[active] < temperature_threshold, temperature_init, temperature_final).astype(np.float32)
        sr = root_search_batch(model, act, temps, sims, c, soft_k, sparse_ply=sparse_ply, sparse_top_k=sparse_top_k)
        leaf_evals += sr["leaf_evals"]
        n = active.size
        S.append(sr["model_input"]); L.append(sr["legal_mask"]); P.append(sr["policy_dense"])
        signs.append(np.where(act["current_player"] >= 0, 1, -1).astype(np.int8))
        rows = np.arange(size, size + n)
        step_index[active, step_counts[active]] = rows
        step_counts[active] += 1
        size += n
        slots, res, _ = O.self_play_step_inplace(states, plies, done, active, sr["chosen_codes"], sr["terminal_mask"],
                                                 sr["chosen_valid"], max_game_plies, soft_k)
        if slots.size:
            soft = O.soft_value_from_board(states["board"][slots], soft_k)
            finished.append((slots.copy(), res.copy(), soft))
        waves += 1
    elapsed = time.perf_counter() - t0
    state_t = np.concatenate(S) if S else np.zeros((0, 11, 6, 6), np.float32)
    sg = np.concatenate(signs) if signs else np.zeros(0, np.int8)
    vt = np.full(size, np.nan, np.float32); svt = np.full(size, np.nan, np.float32)
    outcome = np.zeros(3, np.int64)
    for slots, res, soft in finished:
        _, _, co = O.finalize_trajectory_inplace(vt, svt, sg, step_index, step_counts, slots, res, soft)
        outcome += co
    tensors = dict(state_tensors=state_t, legal_masks=np.concatenate(L) if L else np.zeros((0, 220), bool),
                   policy_targets=np.concatenate(P) if P else np.zeros((0, 220), np.float32),
                   value_targets=vt, soft_value_targets=svt)
    stats = dict(num_positions=size, elapsed_sec=elapsed, positions_per_sec=size / max(elapsed, 1e-9),
                 leaf_evals=leaf_evals, black_wins=int(outcome[0]), white_wins=int(outcome[1]), draws=int(outcome[2]),
                 avg_game_length=float(step_counts.mean()))
    return tensors, stats


# ---------------------------------------------------------------------------------------------
# variant P (full tree, one leaf per game per simulation)
# ---------------------------------------------------------------------------------------------
def tree_search_batch(evaluate: Callable[[Dict[str, np.ndarray]], Tuple[np.ndarray, np.ndarray]], trees, sims: int,
                      noise_fn: Optional[Callable[[int, int], np.ndarray]] = None, eps: float = 0.25) -> int:
    """Run `sims` simulations on every OracleTree (portable_mcts.py:570-650).  `evaluate(states) ->
    (priors[B,220], values[B])`.  Roots kept by `advance` are not re-evaluated; with noise they get a fresh mix
    (:617-621).  Returns the number of leaf evaluations."""
    evals = 0
    pend = [i for i, t in enumerate(trees) if t.prepare_root()]
    if pend:
        st = O.batch_from_states([trees[i].pending_state() for i in pend])
        pri, val = evaluate(st)
        evals += len(pend)
        for k, i in enumerate(pend):
            nz = noise_fn(i, int((O.encode_actions(O.select_states(st, [k]))[0]).sum())) if noise_fn else None
            trees[i].complete(pri[k], float(val[k]), nz, eps)
    if noise_fn:
        fresh = set(pend)
        for i, t in enumerate(trees):
            n = int(t.root_children()[0].size)
            if i not in fresh and n > 0:
                t.root_noise(noise_fn(i, n), eps)
    for _ in range(sims):
        pend = [i for i, t in enumerate(trees) if t.select()]
        if not pend:
            continue
        st = O.batch_from_states([trees[i].pending_state() for i in pend])
        pri, val = evaluate(st)
        evals += len(pend)
        for k, i in enumerate(pend):
            trees[i].complete(pri[k], float(val[k]))
    return evals


def tree_search_waves(evaluate: Callable[[Dict[str, np.ndarray]], Tuple[np.ndarray, np.ndarray]], trees, sims: int,
                      batch_k: int) -> int:
    """The legacy search of src/mcts.py:280-497 (batch_K leaves per wave, no virtual loss) for several trees at once:
    roots expanded without backup, then waves of up to `batch_k` distinct leaves per tree -- terminal leaves are
    backed up at once, the others are evaluated together, expanded and backed up in leaf order.  With batch_k = 1
    this is tree_search_batch.  Returns the number of network evaluations."""
    evals = 0
    fresh = [i for i, t in enumerate(trees) if t.prepare_root()]
    if fresh:
        pri, val = evaluate(O.batch_from_states([trees[i].pending_state() for i in fresh]))
        evals += len(fresh)
        for k, i in enumerate(fresh):
            trees[i].complete(pri[k], float(val[k]))
    done = [0] * len(trees)
    while True:
        owners, states = [], []
        progressed = False
        for i, t in enumerate(trees):
            if done[i] >= sims or t.root_terminal():
                continue
            got = t.select_wave(min(int(batch_k), sims - done[i]))
            if got == 0:
                done[i] = sims                      # nothing collectable (cannot happen on a live tree)
                continue
            progressed = True
            done[i] += got
            ws = t.wave_states()
            if ws:
                owners.append((i, len(ws)))
                states.extend(ws)
        if states:
            pri, val = evaluate(O.batch_from_states(states))
            evals += len(states)
            off = 0
            for i, n in owners:
                trees[i].complete_wave(pri[off:off + n], val[off:off + n])
                off += n
        if not progressed:
            break
    return evals


def make_net_evaluator(model):
    def evaluate(states):
        planes = O.states_to_model_input(states)
        lp1, lp2, lpm, val = _net_eval(model, planes)
        mask, _ = O.encode_actions(states)   # tensor semantics == python on non-terminal reachable states
        probs, _ = O.project_policy(lp1, lp2, lpm, mask)
        return probs, val
    return evaluate


def make_table_evaluator(planes_bits: np.ndarray, priors: np.ndarray, values: np.ndarray):
    """Evaluator that replays recorded network outputs (tests/golden/g10: every position the reference run evaluated,
    keyed by its packed input planes) -- host-independent, unlike re-evaluating the module."""
    table = {planes_bits[i].tobytes(): i for i in range(planes_bits.shape[0])}

    def evaluate(states):
        planes = O.states_to_model_input(states)
        keys = np.packbits(planes.astype(bool).reshape(planes.shape[0], -1), axis=1)
        rows = np.array([table.get(keys[i].tobytes(), -1) for i in range(keys.shape[0])], np.int64)
        known = rows >= 0                # positions the recorded run never evaluated (finished / masked slots): zeros
        pri = np.zeros((rows.size, priors.shape[1]), np.float32); val = np.zeros(rows.size, np.float32)
        pri[known] = priors[rows[known]]; val[known] = values[rows[known]]
        return pri, val
    return evaluate


def deterministic_pick(idx, vis, vs, pr, pl, root_player: int) -> int:
    """portable_mcts.py:208-261: most visits, then Q (atol 1e-6), then prior (atol 1e-8), then lowest index."""
    q = np.where(vis > 0, np.where(pl == root_player, vs, -vs) / np.maximum(vis, 1), 0.0).astype(np.float32)
    cand = np.nonzero(vis == vis.max())[0]
    cand = cand[np.abs(q[cand] - q[cand].max()) <= np.float32(1e-6)]
    cand = cand[np.abs(pr[cand] - pr[cand].max()) <= np.float32(1e-8)]
    return int(idx[cand.min()])


def self_play_tree(model, num_games: int, sims: int, temperature_init: float = 1.0, temperature_final: float = 0.1,
                   temperature_threshold: int = 10, c: float = 1.0, soft_k: float = 2.0, max_game_plies: int = 512,
                   concurrent_games: Optional[int] = None, reuse_tree: bool = True,
                   policy_target_temperature: Optional[float] = None, policy_target_prior_pseudocount: float = 0.0,
                   time_budget_s: Optional[float] = None, collect: bool = False, batch_k: int = 1,
                   evaluate: Optional[Callable] = None):
    """variant-P deterministic self-play (sample_moves=False, no noise): v1/python/portable_self_play.py:82-284.
    The reference keeps the played child's subtree on every move (`advance_root`, :191); `reuse_tree=False`
    rebuilds the tree instead.  Returns a stats dict (+ the 5 trajectory tensors when `collect`)."""
    evaluate = make_net_evaluator(model) if evaluate is None else evaluate     # states -> (priors220, values)
    wave = num_games if concurrent_games is None else max(1, min(int(concurrent_games), int(num_games)))
    rows: Dict[str, list] = dict(S=[], L=[], P=[], V=[], SV=[])
    positions = evals = 0
    outcome = np.zeros(3, np.int64)
    t0 = time.perf_counter()
    out_of_time = False
    for base in range(0, num_games, wave):
        n = min(wave, num_games - base)
        trees = [O.OracleTree(O.state_from_batch(O.initial_states(1), 0), c) for _ in range(n)]
        cur = [O.state_from_batch(O.initial_states(1), 0) for _ in range(n)]
        steps = [[] for _ in range(n)]
        plies = [0] * n
        done = [False] * n
        waves = 0
        while not all(done):
            if time_budget_s is not None and waves > 0 and time.perf_counter() - t0 > time_budget_s:
                out_of_time = True
                break
            act = [i for i in range(n) if not done[i]]
            if int(batch_k) > 1:                    # the legacy search's waves (src/mcts.py batch_K)
                evals += tree_search_waves(evaluate, [trees[i] for i in act], sims, int(batch_k))
            else:
                evals += tree_search_batch(evaluate, [trees[i] for i in act], sims)
            for i in act:
                t = trees[i]
                if t.root_terminal():
                    done[i] = True
                else:
                    idx, vis, vs, pr, pl = t.root_children()
                    temp = temperature_init if plies[i] < temperature_threshold else temperature_final
                    tt = temp if policy_target_temperature is None else float(policy_target_temperature)
                    pol = np.zeros(220, np.float32)
                    pol[idx] = O.policy_from_visits(vis, tt, pr, policy_target_prior_pseudocount)
                    if collect:
                        one = O.batch_from_states([cur[i]])
                        steps[i].append((O.states_to_model_input(one)[0], O.encode_actions(one)[0][0], pol,
                                         1 if int(cur[i].player) >= 0 else -1))
                    pick = deterministic_pick(idx, vis, vs, pr, pl, t.root_player())
                    positions += 1
                    cur[i] = O.apply_index(cur[i], pick)
                    if not (reuse_tree and t.advance(pick)):
                        trees[i] = O.OracleTree(cur[i], c)
                    plies[i] += 1
                    if O.game_status(cur[i]) != 0 or plies[i] >= max_game_plies:
                        done[i] = True
                if done[i]:
                    st = O.game_status(cur[i])
                    res = 1.0 if st == 1 else (-1.0 if st == -1 else 0.0)
                    outcome[0 if res > 0 else (1 if res < 0 else 2)] += 1
                    if collect:
                        soft = float(O.soft_value_from_board(O.batch_from_states([cur[i]])["board"], soft_k)[0])
                        for x, m, p, sg in steps[i]:
                            rows["S"].append(x); rows["L"].append(m); rows["P"].append(p)
                            rows["V"].append(np.float32(res * sg)); rows["SV"].append(np.float32(soft * sg))
            waves += 1
        if out_of_time:
            break
    elapsed = time.perf_counter() - t0
    out = dict(num_positions=positions, elapsed_sec=elapsed, positions_per_sec=positions / max(elapsed, 1e-9),
               leaf_evals=evals, black_wins=int(outcome[0]), white_wins=int(outcome[1]), draws=int(outcome[2]))
    if collect:
        out["tensors"] = dict(
            state_tensors=np.stack(rows["S"]) if rows["S"] else np.zeros((0, 11, 6, 6), np.float32),
            legal_masks=np.stack(rows["L"]) if rows["L"] else np.zeros((0, 220), bool),
            policy_targets=np.stack(rows["P"]) if rows["P"] else np.zeros((0, 220), np.float32),
            value_targets=np.array(rows["V"], np.float32), soft_value_targets=np.array(rows["SV"], np.float32))
    return out
