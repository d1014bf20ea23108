"""`v0_core` operator surface on MI355X.

Python-side mirror of the reference's PyBind11 module (v0/src/bindings/module.cpp:874-1482): same
operator names, argument names/order, dtypes, shapes and return tuples, for the operators the v1
self-play path uses.  Every operator launches hand-written gfx950 kernels through the C ABI of
libliuzhou_hip.so on the *current* torch stream.  Like the reference extension, which dispatches on
`board.device().is_cuda()` (v0/src/game/fast_legal_mask.cpp:453), CPU tensors go to the host build of the same ABI
(libliuzhou_host.so, csrc/lz_host.cpp: plain loops over the same bitboard rules) with the reference's CPU error
convention (an illegal action raises); the search engines and the network kernel are HIP-only.

Two bindings of the same C ABI carry the operators (round 5):
  * `native`  -- the compiled PyBind11 + torch layer `liuzhou_amd/_v0_core_native*.so` (csrc/v0_core_ext.cpp, built by
                 `liuzhou_amd.build.build_ext` / `__graft_entry__.build()`): what the reference's module.cpp is to its
                 kernels; a few microseconds of host time per call.  Used when it is built (LZ_V0_CORE_NATIVE=0 turns it off).
  * `python`  -- the ctypes layer below (12 - 22 us per call); always available, and the only carrier of the helpers that
                 are not part of the reference surface (`self_play_step_raw`, ...).
`binding("native" | "python")` returns either as a namespace (the parity tests run over both); the module-level names
are the native ones when available.

The scalar half of the reference module -- `GameState`, `MoveRecord`, `ActionCode`, `TensorStateBatch`, `Player`,
`ActionType` and the per-state rule functions (`generate_*`, `apply_*`, ...: module.cpp:877-1156) -- is `v0_scalar.py`
over the host library's scalar C ABI (include/liuzhou_scalar.h), re-exported here.

Drop-in use: put `liuzhou_amd/dropin` on PYTHONPATH, then `import v0_core` resolves to this module.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from . import _lib as L


# The scalar half of the module (module.cpp:877-1155): enums, GameState / MoveRecord / ActionCode / TensorStateBatch and the
# one-state rule functions, over include/liuzhou_scalar.h in the host library.  Shared by both operator bindings.
from .v0_scalar import *  # noqa: F401,F403,E402
from .v0_scalar import Phase  # noqa: F401,E402  (v0/include/v0/game_state.hpp, module.cpp:877-885)


def version() -> str:
    return L.lib().lz_version().decode()


def _c(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _state12(board, marks_black, marks_white, phase, current_player, pmr_req, pmr_rem, pcr_req, pcr_rem,
             forced, move_count=None, moves_since_capture=None):
    b = _c(board, torch.int8)
    ts = [b, _c(marks_black, torch.bool), _c(marks_white, torch.bool)]
    for t in (phase, current_player, pmr_req, pmr_rem, pcr_req, pcr_rem, forced):
        ts.append(_c(t, torch.int64))
    ts.append(_c(move_count, torch.int64) if move_count is not None else ts[3])
    ts.append(_c(moves_since_capture, torch.int64) if moves_since_capture is not None else ts[3])
    return ts


def encode_actions_fast(board, marks_black, marks_white, phase, current_player, pending_marks_required,
                        pending_marks_remaining, pending_captures_required, pending_captures_remaining,
                        forced_removals_done, placement_dim: int, movement_dim: int, selection_dim: int,
                        auxiliary_dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """module.cpp:1294-1310 -> (mask bool[B,T], metadata int32[B,T,4])"""
    ts = _state12(board, marks_black, marks_white, phase, current_player, pending_marks_required,
                  pending_marks_remaining, pending_captures_required, pending_captures_remaining,
                  forced_removals_done)
    B = int(ts[0].shape[0])
    T = int(placement_dim + movement_dim + selection_dim + auxiliary_dim)
    mask = torch.empty((B, T), dtype=torch.bool, device=board.device)
    meta = torch.empty((B, T, 4), dtype=torch.int32, device=board.device)
    s = L.soa(ts)
    with L.device_ctx(board.device):
        st = L.lib_for(board).lz_encode_actions_fast(C.byref(s), L.i64(B), L.i64(placement_dim), L.i64(movement_dim),
                                            L.i64(selection_dim), L.i64(auxiliary_dim), L.ptr(mask), L.ptr(meta),
                                            L.stream_ptr(board.device))
    L.check(st, "encode_actions_fast")
    return mask, meta


def _alloc_states(n: int, device) -> list:
    out = [torch.empty((n, 6, 6), dtype=torch.int8, device=device),
           torch.empty((n, 6, 6), dtype=torch.bool, device=device),
           torch.empty((n, 6, 6), dtype=torch.bool, device=device)]
    out += [torch.empty((n,), dtype=torch.int64, device=device) for _ in range(9)]
    return out


def batch_apply_moves(board, marks_black, marks_white, phase, current_player, pending_marks_required,
                      pending_marks_remaining, pending_captures_required, pending_captures_remaining,
                      forced_removals_done, move_count, moves_since_capture, action_codes, parent_indices):
    """module.cpp:1311-1327 -> 12-tuple of child tensors [N,...] (GPU semantics: illegal == no-op)."""
    ts = _state12(board, marks_black, marks_white, phase, current_player, pending_marks_required,
                  pending_marks_remaining, pending_captures_required, pending_captures_remaining,
                  forced_removals_done, move_count, moves_since_capture)
    codes = _c(action_codes.to(board.device), torch.int32)
    parents = _c(parent_indices.to(board.device), torch.int64).view(-1)
    if codes.dim() != 2 or codes.shape[1] != 4:
        raise RuntimeError("action_codes must be (N, 4).")
    N = int(codes.shape[0])
    if int(parents.numel()) != N:
        raise RuntimeError("parent_indices must align with action_codes.")
    out = _alloc_states(N, board.device)
    si, so = L.soa(ts), L.soa(out)
    with L.device_ctx(board.device):
        st = L.lib_for(board).lz_batch_apply_moves(C.byref(si), L.i64(ts[0].shape[0]), L.ptr(codes), L.ptr(parents),
                                          L.i64(N), C.byref(so), L.stream_ptr(board.device))
    L.check(st, "batch_apply_moves")
    return tuple(out)


_STATE_NAMES = ("board", "marks_black", "marks_white", "phase", "current_player", "pending_marks_required",
                "pending_marks_remaining", "pending_captures_required", "pending_captures_remaining",
                "forced_removals_done", "move_count", "moves_since_capture")


def _check_inplace_states(ts, op: str) -> None:
    """The in-place operators pass the caller's storages to the kernels as they are (a converted copy would not be
    mutated): dtype, device, contiguity and batch size are checked instead, as module.cpp's TORCH_CHECKs do."""
    board = ts[0]
    if board.dim() < 1:
        raise RuntimeError(f"{op}: board must be [B, 6, 6]")
    B = int(board.shape[0])
    for i, (name, t) in enumerate(zip(_STATE_NAMES, ts)):
        want = torch.int8 if i == 0 else torch.bool if i < 3 else torch.int64
        if t.dtype != want:
            raise RuntimeError(f"{op}: {name} must be {str(want).replace('torch.', '')} (it is mutated in place)")
        if t.device != board.device:
            raise RuntimeError(f"{op}: {name} is not on the board's device")
        if not t.is_contiguous():
            raise RuntimeError(f"{op}: state tensors must be contiguous (they are mutated)")
        if int(t.numel()) != (B * 36 if i < 3 else B):
            raise RuntimeError(f"{op}: {name} does not hold {B} states")


def batch_apply_moves_inplace(board, marks_black, marks_white, phase, current_player, pending_marks_required,
                              pending_marks_remaining, pending_captures_required, pending_captures_remaining,
                              forced_removals_done, move_count, moves_since_capture, action_codes, slot_indices):
    """fast_apply_moves_cuda.cu:746-917 (state tensors must be contiguous; they are mutated)."""
    ts = [board, marks_black, marks_white, phase, current_player, pending_marks_required, pending_marks_remaining,
          pending_captures_required, pending_captures_remaining, forced_removals_done, move_count,
          moves_since_capture]
    _check_inplace_states(ts, "batch_apply_moves_inplace")
    codes = _c(action_codes.to(board.device), torch.int32)
    slots = _c(slot_indices.to(board.device), torch.int64).view(-1)
    if codes.dim() != 2 or int(codes.shape[1]) != 4 or int(codes.shape[0]) != int(slots.numel()):
        raise RuntimeError("batch_apply_moves_inplace: action_codes must be [N, 4] with one row per slot index")
    s = L.soa(ts)
    with L.device_ctx(board.device):
        st = L.lib_for(board).lz_batch_apply_moves_inplace(C.byref(s), L.i64(board.shape[0]), L.ptr(codes), L.ptr(slots),
                                                  L.i64(slots.numel()), L.stream_ptr(board.device))
    L.check(st, "batch_apply_moves_inplace")


def states_to_model_input(board, marks_black, marks_white, phase, current_player) -> torch.Tensor:
    """module.cpp:1286-1293 -> float32[B,11,6,6]"""
    b = _c(board, torch.int8)
    mb, mw = _c(marks_black, torch.bool), _c(marks_white, torch.bool)
    ph, cp = _c(phase, torch.int64), _c(current_player, torch.int64)
    B = int(b.shape[0])
    out = torch.empty((B, 11, 6, 6), dtype=torch.float32, device=board.device)
    with L.device_ctx(board.device):
        st = L.lib_for(board).lz_states_to_model_input(L.ptr(b), L.ptr(mb), L.ptr(mw), L.ptr(ph), L.ptr(cp), L.i64(B),
                                              L.ptr(out), L.stream_ptr(board.device))
    L.check(st, "states_to_model_input")
    return out


def project_policy_logits_fast(log_p1, log_p2, log_pmc, legal_mask, placement_dim: int, movement_dim: int,
                               selection_dim: int, auxiliary_dim: int):
    """module.cpp:1328-1338 -> (probs, masked_logits) in the heads' dtype (computed in fp32)."""
    if legal_mask.dtype != torch.bool:
        raise RuntimeError("legal_mask must be of dtype bool.")
    if not (log_p1.dtype == log_p2.dtype == log_pmc.dtype):
        raise RuntimeError("All policy heads must share the same dtype.")
    out_dtype = log_p1.dtype
    B = int(log_p1.shape[0])
    p1 = _c(log_p1.reshape(B, -1), torch.float32)
    p2 = _c(log_p2.reshape(B, -1), torch.float32)
    pm = _c(log_pmc.reshape(B, -1), torch.float32)
    T = int(placement_dim + movement_dim + selection_dim + auxiliary_dim)
    if tuple(legal_mask.shape) != (B, T):
        raise RuntimeError(f"legal_mask expected shape ({B}, {T}), got {tuple(legal_mask.shape)}.")
    mk = _c(legal_mask, torch.bool)
    probs = torch.empty((B, T), dtype=torch.float32, device=log_p1.device)
    ml = torch.empty((B, T), dtype=torch.float32, device=log_p1.device)
    with L.device_ctx(log_p1.device):
        st = L.lib_for(log_p1).lz_project_policy_logits_fast(L.ptr(p1), L.ptr(p2), L.ptr(pm), L.ptr(mk), L.i64(B),
                                                   L.i64(placement_dim), L.i64(movement_dim), L.i64(selection_dim),
                                                   L.i64(auxiliary_dim), L.ptr(probs), L.ptr(ml),
                                                   L.stream_ptr(log_p1.device))
    L.check(st, "project_policy_logits_fast")
    if out_dtype != torch.float32:
        return probs.to(out_dtype), ml.to(out_dtype)
    return probs, ml


PACK_CAP = 80   # >= max legal actions of any state (placement 36, movement <= 72)


def root_pack_rows(legal_mask, probs, metadata, cap: int = PACK_CAP):
    """Sync-free fixed-capacity form: (counts i32[B], legal_index i32[B,cap], priors f32[B,cap], codes i32[B,cap,4])."""
    mk = _c(legal_mask, torch.bool)
    pr = _c(probs, torch.float32)
    md = _c(metadata, torch.int32)
    B, T = int(mk.shape[0]), int(mk.shape[1])
    dev = mk.device
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    lidx = torch.empty((B, cap), dtype=torch.int32, device=dev)
    pri = torch.empty((B, cap), dtype=torch.float32, device=dev)
    codes = torch.empty((B, cap, 4), dtype=torch.int32, device=dev)
    with L.device_ctx(dev):
        st = L.lib_for(mk).lz_root_pack_rows(L.ptr(mk), L.ptr(pr), L.ptr(md), L.i64(B), L.i64(T), L.i64(cap),
                                       L.ptr(counts), L.ptr(lidx), L.ptr(pri), L.ptr(codes), L.stream_ptr(dev))
    L.check(st, "root_pack_sparse_actions")
    return counts, lidx, pri, codes


def root_pack_sparse_actions(legal_mask, probs, metadata):
    """module.cpp:1357-1362 -> the reference's 10-tuple with data-dependent [R, Amax] shapes.
    Entirely behind the C ABI (include/liuzhou_hip.h: lz_root_pack_rows -> lz_root_pack_plan -> read {R, Amax, N} ->
    lz_root_pack_fill): three kernels and ONE host read of the sizes (the reference reads R and Amax separately,
    module.cpp:295,310); this wrapper only allocates the outputs."""
    if legal_mask.dim() != 2 or probs.dim() != 2 or metadata.dim() != 3 or metadata.shape[2] != 4:
        raise RuntimeError("legal_mask [B,A], probs [B,A], metadata [B,A,4] expected")
    counts, lidx, pri, codes = root_pack_rows(legal_mask, probs, metadata)
    dev = legal_mask.device
    B = int(counts.shape[0])
    rank = torch.empty((B,), dtype=torch.int32, device=dev)
    child_off = torch.empty((B,), dtype=torch.int64, device=dev)
    sizes = torch.zeros((3,), dtype=torch.int64, device=dev)
    with L.device_ctx(dev):
        L.check(L.lib_for(counts).lz_root_pack_plan(L.ptr(counts), L.i64(B), L.ptr(rank), L.ptr(child_off), L.ptr(sizes),
                                          L.stream_ptr(dev)), "root_pack_sparse_actions")
    R, M, N = (int(v) for v in sizes.tolist())                     # the one host synchronisation
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    terminal_mask = e((B,), torch.bool)
    valid_root_indices, counts_v = e((R,), torch.int64), e((R,), torch.int64)
    if R == 0:
        M = 0
    valid_mask, legal_index_mat = e((R, M), torch.bool), e((R, M), torch.int64)
    priors_mat, action_code_mat = e((R, M), torch.float32), e((R, M, 4), torch.int32)
    pack_flat_idx, action_codes_all, parent_indices_all = e((N,), torch.int64), e((N, 4), torch.int32), e((N,), torch.int64)
    with L.device_ctx(dev):
        L.check(L.lib_for(counts).lz_root_pack_fill(
            L.ptr(counts), L.ptr(lidx), L.ptr(pri), L.ptr(codes), L.ptr(rank), L.ptr(child_off), L.i64(B),
            L.i64(int(lidx.shape[1])), L.i64(R), L.i64(M), L.i64(N), L.ptr(terminal_mask), L.ptr(valid_root_indices),
            L.ptr(counts_v), L.ptr(valid_mask), L.ptr(legal_index_mat), L.ptr(priors_mat), L.ptr(action_code_mat),
            L.ptr(pack_flat_idx), L.ptr(action_codes_all), L.ptr(parent_indices_all), L.stream_ptr(dev)),
            "root_pack_sparse_actions")
    return (terminal_mask, valid_root_indices, counts_v, valid_mask, legal_index_mat, priors_mat,
            action_code_mat, pack_flat_idx, action_codes_all, parent_indices_all)


def root_puct_allocate_visits(priors, leaf_values, valid_mask, num_simulations: int, exploration_weight: float):
    """module.cpp:1349-1356 -> (visits f32[R,A], value_sum f32[R,A], root_values f32[R])"""
    if priors.dim() != 2 or leaf_values.dim() != 2 or valid_mask.dim() != 2:
        raise RuntimeError("priors / leaf_values / valid_mask must be 2D [R, A]")
    if priors.shape != leaf_values.shape or priors.shape != valid_mask.shape:
        raise RuntimeError("priors, leaf_values and valid_mask shape mismatch")
    if int(num_simulations) <= 0:
        raise RuntimeError("num_simulations must be positive")
    p = _c(priors, torch.float32)
    lv = _c(leaf_values, torch.float32)
    vm = _c(valid_mask, torch.bool)
    R, A = int(p.shape[0]), int(p.shape[1])
    dev = p.device
    visits = torch.zeros((R, A), dtype=torch.float32, device=dev)
    vs = torch.zeros((R, A), dtype=torch.float32, device=dev)
    rv = torch.zeros((R,), dtype=torch.float32, device=dev)
    if R == 0 or A == 0:
        return visits, vs, rv
    with L.device_ctx(dev):
        st = L.lib_for(p).lz_root_puct_allocate_visits(L.ptr(p), L.ptr(lv), L.ptr(vm), L.i64(R), L.i64(A),
                                                  L.i64(num_simulations), C.c_float(float(exploration_weight)),
                                                  L.ptr(visits), L.ptr(vs), L.ptr(rv), L.stream_ptr(dev))
    L.check(st, "root_puct_allocate_visits")
    return visits, vs, rv


def root_finalize_from_visits(legal_index_mat, action_code_mat, valid_mask, visits, value_sum, valid_root_indices,
                              batch_size: int, total_action_dim: int, root_temperatures, sample_moves: bool,
                              uniforms=None):
    """module.cpp:1374-1386 -> (policy_dense f32[B,T], chosen_action_indices i64[B], chosen_action_codes
    i32[B,4], chosen_valid_mask bool[B], root_value f32[R]).  sample_moves draws one Philox uniform per
    root on the device (or uses the injected `uniforms` for parity runs) and samples by inverse CDF from
    the log-space policy of mcts_gpu.py:853-898."""
    dev = visits.device
    li = _c(legal_index_mat, torch.int64)
    ac = _c(action_code_mat, torch.int32)
    vm = _c(valid_mask, torch.bool)
    vi = _c(visits, torch.float32)
    vs = _c(value_sum, torch.float32)
    roots = _c(valid_root_indices, torch.int64)
    temps = _c(root_temperatures, torch.float32)
    R = int(li.shape[0])
    M = int(li.shape[1]) if li.dim() == 2 else 0
    B, T = int(batch_size), int(total_action_dim)
    policy = torch.empty((B, T), dtype=torch.float32, device=dev)
    cidx = torch.empty((B,), dtype=torch.int64, device=dev)
    ccodes = torch.empty((B, 4), dtype=torch.int32, device=dev)
    cvalid = torch.empty((B,), dtype=torch.bool, device=dev)
    rv = torch.empty((R,), dtype=torch.float32, device=dev)
    u = None
    if sample_moves and M > 1:
        u = _c(uniforms, torch.float32) if uniforms is not None else torch.rand((R,), dtype=torch.float32, device=dev)
    with L.device_ctx(dev):
        st = L.lib_for(vi).lz_root_finalize_from_visits(L.ptr(li), L.ptr(ac), L.ptr(vm), L.ptr(vi), L.ptr(vs),
                                                  L.ptr(roots), L.i64(R), L.i64(M), L.i64(B), L.i64(T), L.ptr(temps),
                                                  L.ptr(u), L.ptr(policy), L.ptr(cidx), L.ptr(ccodes), L.ptr(cvalid),
                                                  L.ptr(rv), L.stream_ptr(dev))
    L.check(st, "root_finalize_from_visits")
    return policy, cidx, ccodes, cvalid, rv


def self_play_step_raw(state12, plies, done, active_idx, chosen_action_codes, terminal_mask, chosen_valid_mask,
                       max_game_plies: int, soft_value_k: float):
    """Sync-free core of self_play_step_inplace -> (fin_kind i32[A], result f32[A], soft f32[A])."""
    board = state12[0]
    dev = board.device
    _check_inplace_states(list(state12), "self_play_step_inplace")
    for t in (plies, done):
        if not t.is_contiguous():
            raise RuntimeError("self_play_step_inplace: state tensors must be contiguous (they are mutated)")
        if t.device != dev or int(t.numel()) != int(board.shape[0]):
            raise RuntimeError("self_play_step_inplace: plies / done must hold one entry per state on the board's device")
    act = _c(active_idx.to(dev), torch.int64).view(-1)
    codes = _c(chosen_action_codes.to(dev), torch.int32)
    term = _c(terminal_mask.to(dev), torch.bool).view(-1)
    cval = _c(chosen_valid_mask.to(dev), torch.bool).view(-1)
    A = int(act.numel())
    if codes.dim() != 2 or int(codes.shape[0]) != A or int(codes.shape[1]) != 4:
        raise RuntimeError("chosen_action_codes must be [A, 4]")
    if int(term.numel()) != A or int(cval.numel()) != A:
        raise RuntimeError("terminal_mask / chosen_valid_mask batch mismatch")
    kind = torch.empty((A,), dtype=torch.int32, device=dev)
    res = torch.empty((A,), dtype=torch.float32, device=dev)
    soft = torch.empty((A,), dtype=torch.float32, device=dev)
    s = L.soa(state12)
    with L.device_ctx(dev):
        st = L.lib_for(board).lz_self_play_step_inplace(C.byref(s), L.i64(board.shape[0]), L.ptr(plies), L.ptr(done),
                                               L.ptr(act), L.i64(A), L.ptr(codes), L.ptr(term), L.ptr(cval),
                                               L.i64(max_game_plies), C.c_float(float(soft_value_k)), L.ptr(kind),
                                               L.ptr(res), L.ptr(soft), L.stream_ptr(dev))
    L.check(st, "self_play_step_inplace")
    return act, kind, res, soft


def self_play_step_inplace(board, marks_black, marks_white, phase, current_player, pending_marks_required,
                           pending_marks_remaining, pending_captures_required, pending_captures_remaining,
                           forced_removals_done, move_count, moves_since_capture, plies, done, active_idx,
                           chosen_action_codes, terminal_mask, chosen_valid_mask, max_game_plies: int,
                           soft_value_k: float):
    """module.cpp:1387-1409 -> (finished_slots i64[F], result_from_black f32[F], soft f32[F]); mutates the
    state tensors, `plies` and `done`.  Output order = games ended before the move, then games ended by it,
    each in active order (module.cpp:724-741, :838-856)."""
    if int(max_game_plies) <= 0:
        raise RuntimeError("max_game_plies must be positive")
    if done.dtype != torch.bool or plies.dtype != torch.int64:
        raise RuntimeError("plies must be int64 and done must be bool")
    st12 = [board, marks_black, marks_white, phase, current_player, pending_marks_required,
            pending_marks_remaining, pending_captures_required, pending_captures_remaining, forced_removals_done,
            move_count, moves_since_capture]
    act, kind, res, soft = self_play_step_raw(st12, plies, done, active_idx, chosen_action_codes, terminal_mask,
                                              chosen_valid_mask, max_game_plies, soft_value_k)
    if int(act.numel()) == 0:
        e = torch.empty((0,), dtype=torch.float32, device=board.device)
        return torch.empty((0,), dtype=torch.int64, device=board.device), e, e.clone()
    order = torch.cat([torch.nonzero(kind.eq(1)).view(-1), torch.nonzero(kind.eq(2)).view(-1)])
    return act.index_select(0, order), res.index_select(0, order), soft.index_select(0, order)


def finalize_trajectory_inplace(value_targets, soft_value_targets, player_signs, step_index_matrix, step_counts,
                                slots, result_from_black, soft_value_from_black):
    """module.cpp:1410-1420 -> (final_slots, final_counts, counts_out i64[3]); mutates the two targets."""
    dev = value_targets.device
    if not (value_targets.is_contiguous() and soft_value_targets.is_contiguous()):
        raise RuntimeError("target buffers must be contiguous")
    if value_targets.dtype != torch.float32 or soft_value_targets.dtype != torch.float32:
        raise RuntimeError("target buffers must be float32")
    counts_out = torch.zeros((3,), dtype=torch.int64, device=dev)
    sl = _c(slots.to(dev), torch.int64).view(-1)
    F = int(sl.numel())
    empty = torch.empty((0,), dtype=torch.int64, device=dev)
    if F == 0:
        return empty, empty.clone(), counts_out
    res = _c(result_from_black.to(dev), torch.float32).view(-1)
    sft = _c(soft_value_from_black.to(dev), torch.float32).view(-1)
    if int(res.numel()) != F or int(sft.numel()) != F:
        raise RuntimeError("result_from_black / soft_value_from_black must align with slots")
    signs = _c(player_signs, torch.int8)
    sim = _c(step_index_matrix, torch.int64)
    sc = _c(step_counts, torch.int64)
    keep = torch.empty((F,), dtype=torch.bool, device=dev)
    fcounts = torch.empty((F,), dtype=torch.int64, device=dev)
    with L.device_ctx(dev):
        st = L.lib_for(value_targets).lz_finalize_trajectory_inplace(L.ptr(value_targets), L.ptr(soft_value_targets), L.ptr(signs),
                                                    L.ptr(sim), L.ptr(sc), L.i64(sim.shape[0]), L.i64(sim.shape[1]),
                                                    L.ptr(sl), L.ptr(res), L.ptr(sft), L.i64(F), L.ptr(keep),
                                                    L.ptr(fcounts), L.ptr(counts_out), L.stream_ptr(dev))
    L.check(st, "finalize_trajectory_inplace")
    kidx = torch.nonzero(keep).view(-1)
    return sl.index_select(0, kidx), fcounts.index_select(0, kidx), counts_out


def root_sparse_writeback(legal_index_mat, action_code_mat, valid_mask, legal_policy, local_picks, valid_root_indices,
                          batch_size: int, total_action_dim: int):
    """module.cpp:365-439, :1364-1373 -> (policy_dense f32[B,T], chosen_action_indices i64[B] (-1 where no root),
    chosen_action_codes i32[B,4] (-1), chosen_valid_mask bool[B]): the packed per-root policy scattered back to dense rows
    and the picked actions' codes.  A tensor-library composition in the reference too (no kernel of its own); the fused
    `root_finalize_from_visits` is what the hot path uses."""
    if int(batch_size) < 0:
        raise RuntimeError("batch_size must be non-negative")
    if int(total_action_dim) <= 0:
        raise RuntimeError("total_action_dim must be positive")
    if legal_index_mat.dim() != 2 or valid_mask.dim() != 2 or legal_policy.dim() != 2:
        raise RuntimeError("legal_index_mat / valid_mask / legal_policy must be [R, M]")
    if action_code_mat.dim() != 3 or int(action_code_mat.shape[2]) != 4:
        raise RuntimeError("action_code_mat must be [R, M, 4]")
    R, M = int(legal_index_mat.shape[0]), int(legal_index_mat.shape[1])
    if tuple(valid_mask.shape) != (R, M) or tuple(legal_policy.shape) != (R, M) or tuple(action_code_mat.shape[:2]) != (R, M):
        raise RuntimeError("legal_index_mat / valid_mask / legal_policy / action_code_mat shape mismatch")
    if local_picks.dim() != 1 or int(local_picks.shape[0]) != R or valid_root_indices.dim() != 1 or \
            int(valid_root_indices.shape[0]) != R:
        raise RuntimeError("local_picks / valid_root_indices must be [R]")
    dev = legal_index_mat.device
    if any(t.device != dev for t in (action_code_mat, valid_mask, legal_policy, local_picks, valid_root_indices)):
        raise RuntimeError("all tensors must be on the same device")
    idx, codes = _c(legal_index_mat, torch.int64), _c(action_code_mat, torch.int32)
    weights = _c(legal_policy, torch.float32) * _c(valid_mask, torch.bool).to(torch.float32)
    picks, roots = _c(local_picks, torch.int64), _c(valid_root_indices, torch.int64)
    B, T = int(batch_size), int(total_action_dim)
    rows = torch.zeros((R, T), dtype=torch.float32, device=dev).scatter_add_(1, idx, weights)
    policy_dense = torch.zeros((B, T), dtype=torch.float32, device=dev)
    chosen_idx = torch.full((B,), -1, dtype=torch.int64, device=dev)
    chosen_codes = torch.full((B, 4), -1, dtype=torch.int32, device=dev)
    chosen_valid = torch.zeros((B,), dtype=torch.bool, device=dev)
    policy_dense.index_copy_(0, roots, rows)
    chosen_idx.index_copy_(0, roots, idx.gather(1, picks.view(-1, 1)).view(-1))
    chosen_codes.index_copy_(0, roots, codes.gather(1, picks.view(-1, 1, 1).expand(-1, 1, 4)).view(-1, 4))
    chosen_valid.index_fill_(0, roots, True)
    return policy_dense, chosen_idx, chosen_codes, chosen_valid


def postprocess_value_head(raw_values: torch.Tensor) -> torch.Tensor:
    """v0/src/net/encoding.cpp:81-89 (module.cpp:1340-1342): a (..., 3) win/draw/loss head -> P(win) - P(loss), a scalar
    head -> tanh."""
    if raw_values.dim() >= 2 and int(raw_values.shape[-1]) == 3:
        p = torch.softmax(raw_values, dim=-1)
        return p[..., 0] - p[..., 2]
    return torch.tanh(raw_values)


def apply_temperature_scaling(probs: torch.Tensor, temperature: float, dim: int = -1) -> torch.Tensor:
    """v0/src/net/encoding.cpp:91-113 (module.cpp:1344-1348): p ** (1 / T) over the positive entries, renormalised along
    `dim`; T <= 1e-6 returns a copy."""
    probs = probs.contiguous()
    if float(temperature) <= 1e-6:
        return probs.clone()
    d = dim + probs.dim() if dim < 0 else dim
    if not 0 <= d < probs.dim():
        raise RuntimeError("Invalid dimension for temperature scaling")
    powered = torch.where(probs > 0, probs.pow(1.0 / max(float(temperature), 1e-6)), torch.zeros_like(probs))
    sums = powered.sum(d, keepdim=True)
    return torch.where(sums > 0, powered / sums, torch.zeros_like(powered))


# ---- the two bindings ---------------------------------------------------------------------------------------------------
NATIVE_OPS = ("encode_actions_fast", "batch_apply_moves", "batch_apply_moves_inplace", "states_to_model_input",
              "project_policy_logits_fast", "root_pack_rows", "root_pack_sparse_actions", "root_puct_allocate_visits",
              "root_finalize_from_visits", "self_play_step_inplace", "finalize_trajectory_inplace", "root_sparse_writeback",
              "postprocess_value_head", "apply_temperature_scaling")
_python_ops = {name: globals()[name] for name in NATIVE_OPS}
_native = None
_native_error = None


def _load_native():
    """Import liuzhou_amd/_v0_core_native*.so (if built) and hand it the two builds of the C ABI."""
    import importlib.util
    import os
    from .build import HOST_LIB, LIB, build_host, ext_path
    path = ext_path()
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("liuzhou_amd._v0_core_native", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hip = os.environ.get("LZ_HIP_LIB", LIB)
    host = os.environ.get("LZ_HOST_LIB") or (HOST_LIB if os.path.exists(HOST_LIB) else build_host())
    mod.bind_libraries(hip if os.path.exists(hip) else "", host)       # a missing HIP library: HIP tensors raise, no fallback
    return mod


def binding(kind: str = "auto"):
    """The operator surface over one binding: "native" (compiled), "python" (ctypes) or "auto" (what this module exports)."""
    import types
    if kind == "auto":
        kind = "native" if _native is not None else "python"
    if kind == "python":
        return types.SimpleNamespace(kind="python", **_python_ops)
    if kind == "native":
        if _native is None:
            raise RuntimeError(f"the compiled v0_core layer is not available ({_native_error or 'not built: python -m liuzhou_amd.build'})")
        return types.SimpleNamespace(kind="native", **{name: getattr(_native, name) for name in NATIVE_OPS})
    raise ValueError(f"binding must be native / python / auto, got {kind!r}")


def active_binding() -> str:
    return "native" if _native is not None else "python"


def _activate():
    global _native, _native_error
    import os
    if os.environ.get("LZ_V0_CORE_NATIVE", "1").strip() in ("0", "off", "false"):
        _native_error = "switched off by LZ_V0_CORE_NATIVE"
        return
    try:
        _native = _load_native()
    except Exception as exc:      # a stale or broken build must not take the operators away: the ctypes layer stays
        _native, _native_error = None, repr(exc)
        return
    if _native is None:
        _native_error = "not built"
        return
    for name in NATIVE_OPS:
        globals()[name] = getattr(_native, name)


_activate()


def __getattr__(name: str):
    """`MCTSConfig`, `MCTSCore`, `InferenceEngine`, `EvalBatcher`, `TorchScriptRunner` (module.cpp:1158-1284,1422-1479): adapters over the device tree engine,
    resolved lazily (they sit above the modules that import this one)."""
    if name in ("MCTSConfig", "MCTSCore", "InferenceEngine", "EvalBatcher", "TorchScriptRunner"):
        from . import mcts_core
        return getattr(mcts_core, name)
    raise AttributeError(f"module 'v0_core' has no attribute {name!r}")
