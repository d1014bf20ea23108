"""CPU: the bitboard rule header shared with the HIP kernels (liuzhou_amd/csrc/lz_rules.h), compiled for
the host, against the golden vectors.  Catches rule bugs before any GPU minute is spent."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import lz_oracle as O
from tests.golden_utils import load, states, unpack_mask, states_equal

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_check.cpp")
LIB = os.path.join(HERE, "_build", "liblz_hostcheck.so")


@pytest.fixture(scope="module")
def hc():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    deps = [SRC, os.path.join(HERE, "..", "liuzhou_amd", "csrc", "lz_rules.h"),
            os.path.join(HERE, "..", "liuzhou_amd", "csrc", "lz_soa.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC])
    return C.CDLL(LIB)


def _encode(hc, st, ad=4, fallback=1):
    a = O._norm(st)
    B = a["board"].shape[0]
    T = 216 + ad
    mask = np.zeros((B, T), np.uint8); meta = np.zeros((B, T, 4), np.int32)
    v = O._view(a)
    hc.hc_encode_actions(C.byref(v), C.c_int64(B), C.c_int64(ad), C.c_void_p(mask.ctypes.data),
                         C.c_void_p(meta.ctypes.data), C.c_int(fallback))
    return mask.astype(bool), meta


def test_bitboard_legal_masks_reachable(hc):
    z = load("g1_rules.npz")
    mask, meta = _encode(hc, states(z, "s"))
    assert np.array_equal(mask, unpack_mask(z["legal_mask"], 220))
    assert np.array_equal(meta, z["metadata"].astype(np.int32))


@pytest.mark.parametrize("aux,key", [(1, "t217"), (4, "t220")])
def test_bitboard_legal_masks_garbage(hc, aux, key):
    z = load("g3_garbage.npz")
    mask, meta = _encode(hc, states(z, "s"), ad=aux)
    assert np.array_equal(mask, unpack_mask(z[f"mask_{key}"], 216 + aux))
    assert np.array_equal(meta, z[f"meta_{key}"].astype(np.int32))


def test_bitboard_transitions(hc):
    z = load("g1_rules.npz")
    st = O._norm(states(z, "s"))
    parents = np.ascontiguousarray(z["child_parent"], np.int64)
    codes = np.ascontiguousarray(z["metadata"].astype(np.int32)[parents, z["child_action"].astype(np.int64)])
    N = parents.shape[0]
    out = O._norm(O.empty_states(N))
    vi, vo = O._view(st), O._view(out)
    hc.hc_apply_moves(C.byref(vi), C.c_int64(st["board"].shape[0]), C.c_void_p(codes.ctypes.data),
                      C.c_void_p(parents.ctypes.data), C.c_int64(N), C.byref(vo))
    ok, field = states_equal(out, states(z, "c"))
    assert ok, field


def test_bitboard_illegal_actions_match_oracle_noop_semantics(hc):
    """Random (mostly illegal) codes: GPU no-op semantics of fast_apply_moves_cuda.cu via the oracle."""
    z = load("g1_rules.npz")
    st = O._norm(states(z, "s"))
    B = st["board"].shape[0]
    rng = np.random.default_rng(5)
    N = 20000
    parents = rng.integers(0, B, N).astype(np.int64)
    codes = np.stack([rng.integers(0, 10, N), rng.integers(-1, 37, N), rng.integers(-1, 5, N),
                      rng.integers(-1, 36, N)], axis=1).astype(np.int32)
    want = O.apply_moves(st, codes, parents, strict=False)
    out = O._norm(O.empty_states(N))
    vi, vo = O._view(st), O._view(out)
    hc.hc_apply_moves(C.byref(vi), C.c_int64(B), C.c_void_p(codes.ctypes.data), C.c_void_p(parents.ctypes.data),
                      C.c_int64(N), C.byref(vo))
    ok, field = states_equal(out, want)
    assert ok, field


def test_bitboard_status_and_python_counts(hc):
    z = load("g2_edges.npz")
    st = O._norm(states(z, "s"))
    B = st["board"].shape[0]
    status = np.zeros(B, np.int32); nlegal = np.zeros(B, np.int32)
    v = O._view(st)
    hc.hc_status(C.byref(v), C.c_int64(B), C.c_void_p(status.ctypes.data), C.c_void_p(nlegal.ctypes.data))
    assert np.array_equal(status, z["status"].astype(np.int32))
    py = unpack_mask(z["py_legal_mask"], 220)
    running = status == 0
    assert np.array_equal(nlegal[running], py[running].sum(1))
