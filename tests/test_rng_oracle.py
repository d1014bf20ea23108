"""CPU: the per-game counter RNG.  The numpy oracle (oracle/rng_oracle.py) is pinned by the known-answer vectors of the
Philox paper / Random123 (`kat_vectors`, philox4x32-10); the engine header (liuzhou_amd/csrc/lz_rng.h), compiled for the
host, must agree with the oracle bit for bit on raw blocks and uniforms and to float rounding on the Gamma draws."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import rng_oracle as R

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_check.cpp")
LIB = os.path.join(HERE, "_build", "liblz_hostcheck.so")

KAT = [  # Random123 kat_vectors: philox4x32 10 rounds: counter, key -> output
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.fixture(scope="module")
def hc():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    deps = [SRC] + [os.path.join(HERE, "..", "liuzhou_amd", "csrc", h) for h in ("lz_rules.h", "lz_soa.h", "lz_rng.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC])
    return C.CDLL(LIB)


def test_oracle_philox_known_answers():
    for ctr, key, want in KAT:
        got = R.philox4x32_10(np.array([ctr], np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want, (ctr, key)


def test_header_philox_equals_oracle(hc):
    rng = np.random.default_rng(5)
    ctr = rng.integers(0, 1 << 32, size=(4096, 4), dtype=np.uint64).astype(np.uint32)
    ctr[: len(KAT)] = np.array([k[0] for k in KAT], np.uint32)
    for key in ((0, 0), (0xffffffff, 0xffffffff), (0xa4093822, 0x299f31d0), (12345, 0)):
        out = np.zeros_like(ctr)
        hc.hc_philox(C.c_void_p(ctr.ctypes.data), C.c_uint32(key[0]), C.c_uint32(key[1]), C.c_int64(ctr.shape[0]),
                     C.c_void_p(out.ctypes.data))
        assert np.array_equal(out, R.philox4x32_10(ctr, key)), key


def test_header_variates_equal_oracle_and_are_slot_independent(hc):
    seed = 12345 + (7 << 32)
    game = np.array([0, 1, 2, 5, 1 << 33, 4095, 77, 77], np.int64)
    ply = np.array([0, 0, 3, 9, 1, 143, 20, 21], np.int64)
    B, K = game.shape[0], 72
    u = np.zeros(B, np.float32)
    hc.hc_rng_uniform(C.c_uint64(seed), C.c_void_p(game.ctypes.data), C.c_void_p(ply.ctypes.data), C.c_int64(B),
                      C.c_int(R.PURPOSE_PICK), C.c_void_p(u.ctypes.data))
    assert np.array_equal(u, R.uniform(seed, game, ply, R.PURPOSE_PICK))
    assert (u >= 0).all() and (u < 1).all()
    g = np.zeros((B, K), np.float32)
    hc.hc_rng_gamma(C.c_uint64(seed), C.c_void_p(game.ctypes.data), C.c_void_p(ply.ctypes.data), C.c_int64(B),
                    C.c_float(0.3), C.c_int64(K), C.c_void_p(g.ctypes.data))
    np.testing.assert_allclose(g, R.gamma(seed, game, ply, K, 0.3), rtol=2e-5, atol=1e-30)
    # a pure function of (seed, game, ply, index): permuting the batch permutes the rows
    perm = np.array([3, 0, 7, 1, 6, 2, 5, 4])
    g2 = np.zeros((B, K), np.float32)
    gp, pp = np.ascontiguousarray(game[perm]), np.ascontiguousarray(ply[perm])
    hc.hc_rng_gamma(C.c_uint64(seed), C.c_void_p(gp.ctypes.data), C.c_void_p(pp.ctypes.data), C.c_int64(B),
                    C.c_float(0.3), C.c_int64(K), C.c_void_p(g2.ctypes.data))
    assert np.array_equal(g2, g[perm])
    assert not np.array_equal(g[6], g[7])          # same game, next ply: fresh noise


def test_oracle_gamma_distribution():
    """Gamma(0.3): mean 0.3, variance 0.3; Gamma(2.5): mean / variance 2.5 (200 k draws each)."""
    for alpha in (0.3, 2.5):
        x = R.gamma(99, np.arange(4000), 0, 50, alpha).astype(np.float64).reshape(-1)
        assert abs(x.mean() - alpha) < 0.01 * max(1.0, alpha), (alpha, x.mean())
        assert abs(x.var() - alpha) < 0.03 * max(1.0, alpha), (alpha, x.var())
        assert (x > 0).all()
    u = R.uniform(3, np.arange(200000), 1)
    assert abs(float(u.mean()) - 0.5) < 0.003 and abs(float(u.var()) - 1 / 12) < 0.002
