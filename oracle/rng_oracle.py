"""CPU ORACLE of the engine's per-game counter RNG (test infrastructure only; never imported by the product).

Restates, in numpy, the published Philox4x32-10 generator (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
easy as 1, 2, 3", SC'11; Random123 v1.09 `philox.h`) and the variate pipeline of liuzhou_amd/csrc/lz_rng.h.  The
reference draws its noise / samples from torch and numpy library generators (v1/python/mcts_gpu.py:1329-1339,
1410-1424; src/mcts.py:488-491), whose streams cannot be reproduced across frameworks -- SURVEY.md section 7 hard
part 3 -- so this generator is ours; it is pinned by the paper's known-answer vectors (tests/test_rng_oracle.py).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
PURPOSE_NOISE, PURPOSE_PICK, PURPOSE_OPENING = 0, 1, 2


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """ctr uint32[N,4], key (k0, k1) -> uint32[N,4]."""
    c = np.asarray(ctr, np.uint64).reshape(-1, 4).copy()
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = np.stack([hi1 ^ c[:, 1] ^ np.uint64(k0), lo1, hi0 ^ c[:, 3] ^ np.uint64(k1), lo0], axis=1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def draw(seed: int, game, ply, purpose: int, index, attempt) -> np.ndarray:
    """Counter layout of lz_rng.h::draw, broadcast over the array arguments -> uint32[N,4]."""
    game, ply, index, attempt = np.broadcast_arrays(np.asarray(game, np.int64), np.asarray(ply, np.int64),
                                                    np.asarray(index, np.int64), np.asarray(attempt, np.int64))
    g = game.astype(np.uint64).reshape(-1)
    c = np.stack([g & MASK, g >> np.uint64(32), ply.astype(np.uint64).reshape(-1) & MASK,
                  (np.uint64(purpose & 3) | ((index.astype(np.uint64).reshape(-1) & np.uint64(0x3FF)) << np.uint64(2)) |
                   (attempt.astype(np.uint64).reshape(-1) << np.uint64(12))) & MASK], axis=1)
    return philox4x32_10(c, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def u01(x):
    return ((np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def u01_open0(x):
    return (((np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) *
            np.float32(1.0 / 16777216.0)).astype(np.float32)


def uniform(seed: int, game, ply, purpose: int = PURPOSE_PICK) -> np.ndarray:
    return u01(draw(seed, game, ply, purpose, 0, 0)[:, 0])


def gamma(seed: int, game, ply, count: int, alpha: float) -> np.ndarray:
    """float32[B, count]: lz_rng.h::gamma_draw for index 0..count-1 of every (game, ply)."""
    game = np.asarray(game, np.int64).reshape(-1)
    ply = np.broadcast_to(np.asarray(ply, np.int64), game.shape).reshape(-1)
    B = game.shape[0]
    gg = np.repeat(game, count)
    pp = np.repeat(ply, count)
    kk = np.tile(np.arange(count, dtype=np.int64), B)
    f = np.float32
    a = f(alpha + 1.0) if alpha < 1.0 else f(alpha)
    d = f(a - f(1.0) / f(3.0))
    c = f(f(1.0) / np.sqrt(f(9.0) * d, dtype=np.float32))
    out = np.zeros(B * count, np.float32)
    boost = np.ones(B * count, np.float32)
    todo = np.ones(B * count, bool)
    for attempt in range(64):
        idx = np.nonzero(todo)[0]
        if idx.size == 0:
            break
        r = draw(seed, gg[idx], pp[idx], PURPOSE_NOISE, kk[idx], attempt)
        if attempt == 0 and alpha < 1.0:
            boost[idx] = np.exp(np.log(u01_open0(r[:, 3])) / f(alpha)).astype(np.float32)
        n = (np.sqrt(f(-2.0) * np.log(u01_open0(r[:, 0]))) * np.cos(f(6.283185307179586) * u01(r[:, 1]))).astype(np.float32)
        t = (f(1.0) + c * n).astype(np.float32)
        ok_t = t > 0
        v = (t * t * t).astype(np.float32)
        u = u01_open0(r[:, 2])
        with np.errstate(invalid="ignore", divide="ignore"):
            acc = ok_t & (np.log(u) < (f(0.5) * n * n + d - d * v + d * np.log(np.where(ok_t, v, f(1.0)))).astype(np.float32))
        g = (d * v * boost[idx]).astype(np.float32)
        out[idx[acc]] = np.maximum(g[acc], f(1e-30))
        todo[idx[acc]] = False
    rest = np.nonzero(todo)[0]
    out[rest] = np.maximum(d * boost[rest], f(1e-30))
    return out.reshape(B, count)
