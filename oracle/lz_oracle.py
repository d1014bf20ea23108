"""ctypes + numpy front-end of the CPU ORACLE (oracle/lz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by liuzhou_amd/ (the product path).

The C part restates the rule engine / encodings / searches; the data-dependent-shape host ops of
the reference's PyBind module (pack / finalize / self-play step / trajectory finalize) are
restated here in numpy.  Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblz_oracle.so")

STATE_FIELDS = (
    "board", "marks_black", "marks_white", "phase", "current_player",
    "pending_marks_required", "pending_marks_remaining",
    "pending_captures_required", "pending_captures_remaining",
    "forced_removals_done", "move_count", "moves_since_capture",
)
PLACEMENT_DIM, MOVEMENT_DIM, SELECTION_DIM, AUXILIARY_DIM = 36, 144, 36, 4
TOTAL_DIM = 220
MAX_MOVE_COUNT, NO_CAPTURE_DRAW_LIMIT, LOSE_PIECE_THRESHOLD = 144, 36, 4


def build(force: bool = False) -> str:
    """Compile liblz_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "lz_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Batch(C.Structure):
    _fields_ = [
        ("board", C.c_void_p), ("marks_black", C.c_void_p), ("marks_white", C.c_void_p),
        ("phase", C.c_void_p), ("current_player", C.c_void_p),
        ("pending_marks_required", C.c_void_p), ("pending_marks_remaining", C.c_void_p),
        ("pending_captures_required", C.c_void_p), ("pending_captures_remaining", C.c_void_p),
        ("forced_removals_done", C.c_void_p), ("move_count", C.c_void_p),
        ("moves_since_capture", C.c_void_p),
    ]


class CState(C.Structure):
    _fields_ = [
        ("board", C.c_int8 * 36), ("mb", C.c_uint8 * 36), ("mw", C.c_uint8 * 36),
        ("phase", C.c_int64), ("player", C.c_int64),
        ("pm_req", C.c_int64), ("pm_rem", C.c_int64), ("pc_req", C.c_int64), ("pc_rem", C.c_int64),
        ("forced", C.c_int64), ("move_count", C.c_int64), ("msc", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.lzo_apply_moves.restype = C.c_int64
        L.lzo_tree_new.restype = C.c_void_p
        L.lzo_tree_new.argtypes = [C.POINTER(CState), C.c_double]
        L.lzo_tree_free.argtypes = [C.c_void_p]
        L.lzo_tree_prepare_root.argtypes = [C.c_void_p]
        L.lzo_tree_select.argtypes = [C.c_void_p]
        L.lzo_tree_pending_state.argtypes = [C.c_void_p, C.POINTER(CState)]
        L.lzo_tree_complete.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float]
        L.lzo_tree_root_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.lzo_tree_select_wave.argtypes = [C.c_void_p, C.c_int]
        L.lzo_set_max_backtrack.argtypes = [C.c_int]
        L.lzo_tree_wave_count.argtypes = [C.c_void_p]
        L.lzo_tree_wave_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(CState)]
        L.lzo_tree_complete_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lzo_tree_root_terminal.argtypes = [C.c_void_p]
        L.lzo_tree_root_children.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.lzo_tree_root_visits.argtypes = [C.c_void_p]
        L.lzo_tree_root_value_sum.argtypes = [C.c_void_p]
        L.lzo_tree_root_value_sum.restype = C.c_double
        L.lzo_tree_root_player.argtypes = [C.c_void_p]
        L.lzo_tree_advance.argtypes = [C.c_void_p, C.c_int]
        L.lzo_tree_node_count.argtypes = [C.c_void_p]
        L.lzo_game_status.argtypes = [C.POINTER(CState)]
        L.lzo_legal_indices_py.argtypes = [C.POINTER(CState), C.c_void_p]
        L.lzo_apply_index.argtypes = [C.POINTER(CState), C.c_int, C.POINTER(CState)]
        _lib = L
    return _lib


# ----------------------------------------------------------------------------------------------
# State batches as dicts of numpy arrays (same 12 fields / dtypes as the reference tensors)
# ----------------------------------------------------------------------------------------------
def empty_states(n: int) -> Dict[str, np.ndarray]:
    s = {
        "board": np.zeros((n, 6, 6), np.int8),
        "marks_black": np.zeros((n, 6, 6), np.bool_),
        "marks_white": np.zeros((n, 6, 6), np.bool_),
    }
    for f in STATE_FIELDS[3:]:
        s[f] = np.zeros((n,), np.int64)
    return s


def initial_states(n: int) -> Dict[str, np.ndarray]:
    """v1/python/mcts_gpu.py:123-145"""
    s = empty_states(n)
    s["phase"][:] = 1
    s["current_player"][:] = 1
    return s


def _norm(states: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    out = {}
    out["board"] = np.ascontiguousarray(states["board"], dtype=np.int8)
    out["marks_black"] = np.ascontiguousarray(states["marks_black"]).astype(np.uint8)
    out["marks_white"] = np.ascontiguousarray(states["marks_white"]).astype(np.uint8)
    for f in STATE_FIELDS[3:]:
        out[f] = np.ascontiguousarray(states[f], dtype=np.int64)
    return out


def _view(arrs: Dict[str, np.ndarray]) -> _Batch:
    b = _Batch()
    for f in STATE_FIELDS:
        setattr(b, f, arrs[f].ctypes.data)
    return b


def select_states(states: Dict[str, np.ndarray], idx) -> Dict[str, np.ndarray]:
    return {f: np.ascontiguousarray(states[f][idx]) for f in STATE_FIELDS}


def concat_states(parts: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    return {f: np.concatenate([p[f] for p in parts], axis=0) for f in STATE_FIELDS}


def state_from_batch(states: Dict[str, np.ndarray], i: int) -> CState:
    s = CState()
    s.board[:] = states["board"][i].reshape(-1).astype(np.int8).tolist()
    s.mb[:] = states["marks_black"][i].reshape(-1).astype(np.uint8).tolist()
    s.mw[:] = states["marks_white"][i].reshape(-1).astype(np.uint8).tolist()
    s.phase = int(states["phase"][i]); s.player = int(states["current_player"][i])
    s.pm_req = int(states["pending_marks_required"][i]); s.pm_rem = int(states["pending_marks_remaining"][i])
    s.pc_req = int(states["pending_captures_required"][i]); s.pc_rem = int(states["pending_captures_remaining"][i])
    s.forced = int(states["forced_removals_done"][i]); s.move_count = int(states["move_count"][i])
    s.msc = int(states["moves_since_capture"][i])
    return s


def batch_from_states(cs: Sequence[CState]) -> Dict[str, np.ndarray]:
    out = empty_states(len(cs))
    for i, s in enumerate(cs):
        out["board"][i] = np.array(s.board[:], np.int8).reshape(6, 6)
        out["marks_black"][i] = np.array(s.mb[:], np.uint8).reshape(6, 6).astype(bool)
        out["marks_white"][i] = np.array(s.mw[:], np.uint8).reshape(6, 6).astype(bool)
        out["phase"][i] = s.phase; out["current_player"][i] = s.player
        out["pending_marks_required"][i] = s.pm_req; out["pending_marks_remaining"][i] = s.pm_rem
        out["pending_captures_required"][i] = s.pc_req; out["pending_captures_remaining"][i] = s.pc_rem
        out["forced_removals_done"][i] = s.forced; out["move_count"][i] = s.move_count
        out["moves_since_capture"][i] = s.msc
    return out


# ----------------------------------------------------------------------------------------------
# C-backed ops
# ----------------------------------------------------------------------------------------------
def encode_actions(states, pd=36, md=144, sd=36, ad=4) -> Tuple[np.ndarray, np.ndarray]:
    """v0/src/game/fast_legal_mask.cpp:253-418 -> (mask bool[B,T], metadata int32[B,T,4])"""
    a = _norm(states)
    B = a["board"].shape[0]
    T = pd + md + sd + ad
    mask = np.zeros((B, T), np.uint8)
    meta = np.full((B, T, 4), -1, np.int32)
    v = _view(a)
    lib().lzo_encode_actions(C.byref(v), C.c_int64(B), C.c_int64(pd), C.c_int64(md), C.c_int64(sd),
                             C.c_int64(ad), C.c_void_p(mask.ctypes.data), C.c_void_p(meta.ctypes.data))
    return mask.astype(bool), meta


def apply_moves(states, codes, parents, strict: bool = True) -> Dict[str, np.ndarray]:
    """v0/src/game/fast_apply_moves.cpp:595-938 (strict) / _cuda.cu:548-744 (strict=False)"""
    a = _norm(states)
    B = a["board"].shape[0]
    codes = np.ascontiguousarray(codes, np.int32).reshape(-1, 4)
    parents = np.ascontiguousarray(parents, np.int64).reshape(-1)
    N = codes.shape[0]
    out = _norm(empty_states(N))
    vi, vo = _view(a), _view(out)
    rc = lib().lzo_apply_moves(C.byref(vi), C.c_int64(B), C.c_void_p(codes.ctypes.data),
                               C.c_void_p(parents.ctypes.data), C.c_int64(N), C.byref(vo),
                               C.c_int(1 if strict else 0))
    if rc != 0:
        raise RuntimeError(f"illegal action at row {-int(rc) - 1}")
    out["marks_black"] = out["marks_black"].astype(bool)
    out["marks_white"] = out["marks_white"].astype(bool)
    out["board"] = out["board"].reshape(N, 6, 6)
    out["marks_black"] = out["marks_black"].reshape(N, 6, 6)
    out["marks_white"] = out["marks_white"].reshape(N, 6, 6)
    return out


def states_to_model_input(states) -> np.ndarray:
    a = _norm(states)
    B = a["board"].shape[0]
    out = np.zeros((B, 11, 6, 6), np.float32)
    v = _view(a)
    lib().lzo_states_to_model_input(C.byref(v), C.c_int64(B), C.c_void_p(out.ctypes.data))
    return out


def project_policy(lp1, lp2, lpmc, mask, pd=36, md=144, sd=36, ad=4):
    lp1 = np.ascontiguousarray(lp1, np.float32); lp2 = np.ascontiguousarray(lp2, np.float32)
    lpmc = np.ascontiguousarray(lpmc, np.float32)
    m = np.ascontiguousarray(mask).astype(np.uint8)
    B = lp1.shape[0]
    T = pd + md + sd + ad
    probs = np.zeros((B, T), np.float32)
    ml = np.zeros((B, T), np.float32)
    lib().lzo_project_policy(C.c_void_p(lp1.ctypes.data), C.c_void_p(lp2.ctypes.data),
                             C.c_void_p(lpmc.ctypes.data), C.c_void_p(m.ctypes.data), C.c_int64(B),
                             C.c_int64(pd), C.c_int64(md), C.c_int64(sd), C.c_int64(ad),
                             C.c_void_p(probs.ctypes.data), C.c_void_p(ml.ctypes.data))
    return probs, ml


def root_puct(priors, leaf, valid, sims: int, c: float):
    priors = np.ascontiguousarray(priors, np.float32); leaf = np.ascontiguousarray(leaf, np.float32)
    valid = np.ascontiguousarray(valid).astype(np.uint8)
    R, A = priors.shape
    visits = np.zeros((R, A), np.float32); vs = np.zeros((R, A), np.float32)
    rv = np.zeros((R,), np.float32)
    lib().lzo_root_puct(C.c_void_p(priors.ctypes.data), C.c_void_p(leaf.ctypes.data),
                        C.c_void_p(valid.ctypes.data), C.c_int64(R), C.c_int64(A), C.c_int64(sims),
                        C.c_float(c), C.c_void_p(visits.ctypes.data), C.c_void_p(vs.ctypes.data),
                        C.c_void_p(rv.ctypes.data))
    return visits, vs, rv


def game_status(cs: CState) -> int:
    return int(lib().lzo_game_status(C.byref(cs)))


def legal_indices_py(cs: CState) -> List[int]:
    buf = (C.c_int * 80)()
    n = lib().lzo_legal_indices_py(C.byref(cs), buf)
    return [int(buf[i]) for i in range(n)]


def apply_index(cs: CState, a: int) -> CState:
    out = CState()
    rc = lib().lzo_apply_index(C.byref(cs), int(a), C.byref(out))
    if rc != 0:
        raise ValueError(f"illegal action index {a}")
    return out


# ----------------------------------------------------------------------------------------------
# numpy restatements of the reference's data-dependent host ops (v0/src/bindings/module.cpp)
# ----------------------------------------------------------------------------------------------
def root_pack_sparse_actions(legal_mask, probs, metadata):
    """module.cpp:247-363"""
    lm = np.asarray(legal_mask, bool); pr = np.asarray(probs, np.float32); md = np.asarray(metadata, np.int32)
    B, T = lm.shape
    row_counts = lm.sum(1).astype(np.int64)
    terminal = row_counts == 0
    roots = np.nonzero(~terminal)[0].astype(np.int64)
    counts = row_counts[roots]
    if roots.size == 0:
        z = np.zeros
        return (terminal, roots, counts, z((0, 0), bool), z((0, 0), np.int64), z((0, 0), np.float32),
                z((0, 0, 4), np.int32), z((0,), np.int64), z((0, 4), np.int32), z((0,), np.int64))
    R, M = roots.size, int(counts.max())
    valid = np.zeros((R, M), bool); lidx = np.zeros((R, M), np.int64)
    pri = np.zeros((R, M), np.float32); codes = np.zeros((R, M, 4), np.int32)
    flat, codes_all, parents = [], [], []
    for r, b in enumerate(roots):
        idx = np.nonzero(lm[b])[0]
        k = idx.size
        valid[r, :k] = True; lidx[r, :k] = idx; pri[r, :k] = pr[b, idx]; codes[r, :k] = md[b, idx]
        flat.append(r * M + np.arange(k)); codes_all.append(md[b, idx]); parents.append(np.full(k, b))
    denom = np.maximum(pri.sum(1, keepdims=True, dtype=np.float32), np.float32(1e-8))
    pri = (pri / denom).astype(np.float32)
    return (terminal, roots, counts, valid, lidx, pri, codes,
            np.concatenate(flat).astype(np.int64), np.concatenate(codes_all).astype(np.int32),
            np.concatenate(parents).astype(np.int64))


def root_finalize_from_visits(lidx, codes, valid, visits, value_sum, roots, batch_size, total_dim,
                              temps, pick: Optional[np.ndarray] = None):
    """module.cpp:441-535 with sample_moves=False (argmax); `pick` overrides the local picks."""
    visits = np.asarray(visits, np.float32); valid = np.asarray(valid, bool)
    t = np.maximum(np.asarray(temps, np.float32), np.float32(1e-6)).reshape(-1, 1)
    pol = np.power(np.maximum(visits, np.float32(1e-8)), (np.float32(1.0) / t)).astype(np.float32)
    pol = pol * valid.astype(np.float32)
    pol = pol / np.maximum(pol.sum(1, keepdims=True, dtype=np.float32), np.float32(1e-8))
    local = np.argmax(pol, axis=1) if pick is None else np.asarray(pick, np.int64)
    rv = np.asarray(value_sum, np.float32).sum(1, dtype=np.float32) / np.maximum(visits.sum(1, dtype=np.float32), np.float32(1.0))
    R = lidx.shape[0]
    policy_dense = np.zeros((batch_size, total_dim), np.float32)
    chosen_idx = np.full((batch_size,), -1, np.int64)
    chosen_codes = np.full((batch_size, 4), -1, np.int32)
    chosen_valid = np.zeros((batch_size,), bool)
    for r in range(R):
        b = int(roots[r])
        np.add.at(policy_dense[b], lidx[r], pol[r] * valid[r])
        chosen_idx[b] = lidx[r, local[r]]; chosen_codes[b] = codes[r, local[r]]; chosen_valid[b] = True
    return policy_dense, chosen_idx, chosen_codes, chosen_valid, rv.astype(np.float32)


def soft_value_from_board(board, k: float) -> np.ndarray:
    """module.cpp:537-545 / mcts_gpu.py:677-686"""
    b = np.asarray(board).reshape(-1, 36)
    black = (b == 1).sum(1).astype(np.float32); white = (b == -1).sum(1).astype(np.float32)
    return np.tanh(((black - white) / np.float32(18.0)) * np.float32(k)).astype(np.float32)


def terminal_mask_from_next_state(states) -> np.ndarray:
    """mcts_gpu.py:658-675"""
    ph = states["phase"]
    post = (ph == 4) | (ph == 5) | (ph == 7)
    b = np.asarray(states["board"]).reshape(-1, 36)
    bc = (b == 1).sum(1); wc = (b == -1).sum(1)
    win = post & ((bc < LOSE_PIECE_THRESHOLD) | (wc < LOSE_PIECE_THRESHOLD))
    draw = (states["move_count"] >= MAX_MOVE_COUNT) | (states["moves_since_capture"] >= NO_CAPTURE_DRAW_LIMIT)
    return win | draw


def self_play_step_inplace(states, plies, done, active_idx, chosen_codes, terminal_mask,
                           chosen_valid, max_game_plies: int, soft_value_k: float):
    """module.cpp:632-871.  Mutates `states`, `plies`, `done` (numpy arrays) in place."""
    active_idx = np.asarray(active_idx, np.int64).reshape(-1)
    chosen_codes = np.asarray(chosen_codes, np.int32).reshape(-1, 4)
    terminal_mask = np.asarray(terminal_mask, bool).reshape(-1)
    chosen_valid = np.asarray(chosen_valid, bool).reshape(-1)
    slots_out, res_out, soft_out = [], [], []
    if active_idx.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32)
    immediate = terminal_mask | ~chosen_valid
    imm_local = np.nonzero(immediate)[0]
    if imm_local.size:
        slots = active_idx[imm_local]
        done[slots] = True
        player = states["current_player"][slots].astype(np.float32)
        res = np.where(terminal_mask[imm_local], -player, np.float32(0.0)).astype(np.float32)
        slots_out.append(slots); res_out.append(res)
        soft_out.append(soft_value_from_board(states["board"][slots], soft_value_k))
    val_local = np.nonzero(~immediate)[0]
    if val_local.size:
        slots = active_idx[val_local]
        nxt = apply_moves(states, chosen_codes[val_local], slots, strict=True)
        for f in STATE_FIELDS:
            states[f][slots] = nxt[f]
        plies[slots] += 1
        ph = nxt["phase"]
        post = (ph == 4) | (ph == 5) | (ph == 7)
        b = nxt["board"].reshape(-1, 36)
        bc = (b == 1).sum(1); wc = (b == -1).sum(1)
        winner = np.zeros(slots.size, np.int8)
        winner = np.where(post & (bc < LOSE_PIECE_THRESHOLD), np.int8(-1), winner)
        winner = np.where(post & (wc < LOSE_PIECE_THRESHOLD), np.int8(1), winner)
        draw = (nxt["move_count"] >= MAX_MOVE_COUNT) | (nxt["moves_since_capture"] >= NO_CAPTURE_DRAW_LIMIT)
        cap = plies[slots] >= max_game_plies
        fin = np.nonzero((winner != 0) | draw | cap)[0]
        if fin.size:
            fslots = slots[fin]
            done[fslots] = True
            slots_out.append(fslots); res_out.append(winner[fin].astype(np.float32))
            soft_out.append(soft_value_from_board(nxt["board"][fin], soft_value_k))
    if not slots_out:
        return np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32)
    return np.concatenate(slots_out), np.concatenate(res_out), np.concatenate(soft_out)


def finalize_trajectory_inplace(value_targets, soft_targets, player_signs, step_index_matrix,
                                step_counts, slots, result_from_black, soft_from_black):
    """module.cpp:547-630.  Mutates value_targets / soft_targets."""
    slots = np.asarray(slots, np.int64).reshape(-1)
    res = np.asarray(result_from_black, np.float32).reshape(-1)
    soft = np.asarray(soft_from_black, np.float32).reshape(-1)
    counts_out = np.zeros(3, np.int64)
    if slots.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), counts_out
    counts = np.asarray(step_counts, np.int64)[slots]
    keep = np.nonzero(counts > 0)[0]
    if keep.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), counts_out
    fs, fc, fr, fsoft = slots[keep], counts[keep], res[keep], soft[keep]
    counts_out[:] = [(fr > 0).sum(), (fr < 0).sum(), (fr == 0).sum()]
    for g, n, r, s in zip(fs, fc, fr, fsoft):
        idx = np.asarray(step_index_matrix)[g, :n]
        sg = np.asarray(player_signs)[idx].astype(np.float32)
        value_targets[idx] = sg * r
        soft_targets[idx] = sg * s
    return fs, fc, counts_out


# ----------------------------------------------------------------------------------------------
# Variant-P tree (split phase, like v1/python/portable_cpp_mcts.py:270-282)
# ----------------------------------------------------------------------------------------------
class OracleTree:
    def __init__(self, cs: CState, exploration_weight: float = 1.0):
        self._L = lib()
        self._t = self._L.lzo_tree_new(C.byref(cs), C.c_double(exploration_weight))

    def __del__(self):
        try:
            self._L.lzo_tree_free(self._t)
        except Exception:
            pass

    def prepare_root(self) -> bool:
        return bool(self._L.lzo_tree_prepare_root(self._t))

    def select(self) -> bool:
        return bool(self._L.lzo_tree_select(self._t))

    def pending_state(self) -> CState:
        s = CState()
        self._L.lzo_tree_pending_state(self._t, C.byref(s))
        return s

    def complete(self, priors220: np.ndarray, value: float, noise: Optional[np.ndarray] = None,
                 epsilon: float = 0.25) -> None:
        p = np.ascontiguousarray(priors220, np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
        self._L.lzo_tree_complete(self._t, p.ctypes.data, C.c_float(float(value)),
                                  None if nz is None else nz.ctypes.data, C.c_float(float(epsilon)))

    # ---- legacy batch_K waves (src/mcts.py:318-497) ----
    def select_wave(self, to_collect: int) -> int:
        """Collect up to `to_collect` distinct leaves, back the terminal ones up; returns simulations consumed."""
        return int(self._L.lzo_tree_select_wave(self._t, int(to_collect)))

    def wave_states(self):
        out = []
        for j in range(int(self._L.lzo_tree_wave_count(self._t))):
            s = CState()
            self._L.lzo_tree_wave_state(self._t, j, C.byref(s))
            out.append(s)
        return out

    def complete_wave(self, priors220: np.ndarray, values: np.ndarray) -> None:
        p = np.ascontiguousarray(priors220, np.float32).reshape(-1, 220)
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        self._L.lzo_tree_complete_wave(self._t, p.ctypes.data, v.ctypes.data)

    def root_noise(self, noise: np.ndarray, epsilon: float) -> None:
        nz = np.ascontiguousarray(noise, np.float32)
        self._L.lzo_tree_root_noise(self._t, nz.ctypes.data, C.c_float(float(epsilon)))

    def root_terminal(self) -> bool:
        return bool(self._L.lzo_tree_root_terminal(self._t))

    def root_children(self):
        idx = np.zeros(80, np.int32); vis = np.zeros(80, np.int32); vs = np.zeros(80, np.float64)
        pr = np.zeros(80, np.float32); pl = np.zeros(80, np.int32)
        n = self._L.lzo_tree_root_children(self._t, idx.ctypes.data, vis.ctypes.data, vs.ctypes.data,
                                           pr.ctypes.data, pl.ctypes.data)
        return idx[:n].copy(), vis[:n].copy(), vs[:n].copy(), pr[:n].copy(), pl[:n].copy()

    def root_visits(self) -> int:
        return int(self._L.lzo_tree_root_visits(self._t))

    def root_value_sum(self) -> float:
        return float(self._L.lzo_tree_root_value_sum(self._t))

    def root_player(self) -> int:
        return int(self._L.lzo_tree_root_player(self._t))

    def advance(self, action_index: int) -> bool:
        return bool(self._L.lzo_tree_advance(self._t, int(action_index)))

    def node_count(self) -> int:
        return int(self._L.lzo_tree_node_count(self._t))


def policy_from_visits(visits: np.ndarray, temperature: float, priors: Optional[np.ndarray] = None,
                       prior_pseudocount: float = 0.0) -> np.ndarray:
    """portable_mcts.py:150-205 (fp32 like torch): scores = visits (+ beta * normalised priors)."""
    v = np.asarray(visits, np.float32)
    if prior_pseudocount > 0.0:
        p = np.maximum(np.asarray(priors, np.float32), np.float32(1e-8))
        ps = float(p.sum(dtype=np.float32))
        p = np.full_like(p, 1.0 / max(1, p.size)) if (not np.isfinite(ps) or ps <= 0.0) else (p / np.float32(ps))
        v = (v + np.float32(prior_pseudocount) * p).astype(np.float32)
    if temperature <= 1e-6:
        out = np.zeros_like(v); out[int(np.argmax(v))] = 1.0
        return out
    logits = np.full_like(v, -np.inf)
    pos = v > 0
    logits[pos] = np.log(v[pos]) / np.float32(max(temperature, 1e-6))
    m = logits.max()
    e = np.exp(logits - m).astype(np.float32)
    return (e / e.sum(dtype=np.float32)).astype(np.float32)
