// lz_rng.h -- per-game counter RNG of the self-play engine: Philox4x32-10 (Salmon et al., SC'11, "Parallel random
// numbers: as easy as 1, 2, 3"), keyed by the run seed and indexed by (game id, ply, purpose, draw).
//
// The reference draws its Dirichlet root noise and move samples from library generators whose streams depend on the
// batch composition (torch.distributions.Gamma / torch.multinomial on the device generator,
// v1/python/mcts_gpu.py:1329-1339,1410-1424; np.random.dirichlet in src/mcts.py:488-491).  Here every variate is a
// pure function of (seed, game id, ply, purpose, index): a game plays the same moves whichever slot, batch split,
// stream or rank it runs in ("game RNG = Philox(seed, subsequence = game id)").
//
// Counter layout (4 x 32 bit):  c0 = game id low, c1 = game id high, c2 = ply, c3 = purpose | index << 2 | attempt << 12
// Key (2 x 32 bit):             seed low, seed high
// Compiles for the host as well (tests/host_check.cpp, oracle cross-check).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define LZ_RNG_HD __host__ __device__ __forceinline__
#else
#define LZ_RNG_HD inline
#endif

namespace lzrng {

enum Purpose : uint32_t { kPurposeNoise = 0, kPurposePick = 1, kPurposeOpening = 2 };

struct U4 { uint32_t x, y, z, w; };

LZ_RNG_HD void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

LZ_RNG_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(M0, c.x, hi0, lo0);
        mulhilo(M1, c.z, hi1, lo1);
        const U4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

LZ_RNG_HD U4 draw(uint64_t seed, int64_t game, int64_t ply, uint32_t purpose, uint32_t index, uint32_t attempt) {
    const U4 c = {(uint32_t)((uint64_t)game & 0xFFFFFFFFull), (uint32_t)((uint64_t)game >> 32), (uint32_t)ply,
                  (purpose & 3u) | ((index & 0x3FFu) << 2) | (attempt << 12)};
    return philox4x32_10(c, (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32));
}

// 24 random bits -> [0, 1): exact in fp32, never 1
LZ_RNG_HD float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// (0, 1]: safe under log
LZ_RNG_HD float u01_open0(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }

// Gamma(alpha, 1) draw, Marsaglia & Tsang (ACM TOMS 26(3), 2000) on alpha + 1, boosted by U^(1/alpha) for alpha < 1.
// One Philox block per attempt: x, y -> a standard normal (Box-Muller), z -> the acceptance uniform, w (attempt 0)
// -> the boost uniform.  A Dirichlet(alpha) vector is these draws divided by their sum; the expand kernel
// renormalises after mixing, so the unnormalised Gammas are what it consumes.
LZ_RNG_HD float gamma_draw(uint64_t seed, int64_t game, int64_t ply, uint32_t index, float alpha) {
    const float a = alpha < 1.0f ? alpha + 1.0f : alpha;
    const float d = a - 1.0f / 3.0f;
    const float c = 1.0f / sqrtf(9.0f * d);
    float boost = 1.0f;
    for (uint32_t attempt = 0; attempt < 64u; ++attempt) {
        const U4 r = draw(seed, game, ply, kPurposeNoise, index, attempt);
        if (attempt == 0 && alpha < 1.0f) boost = expf(logf(u01_open0(r.w)) / alpha);
        const float n = sqrtf(-2.0f * logf(u01_open0(r.x))) * cosf(6.283185307179586f * u01(r.y));
        const float t = 1.0f + c * n;
        if (!(t > 0.0f)) continue;
        const float v = t * t * t;
        const float u = u01_open0(r.z);
        if (logf(u) < 0.5f * n * n + d - d * v + d * logf(v)) {
            const float g = d * v * boost;
            return g > 1e-30f ? g : 1e-30f;
        }
    }
    return d * boost > 1e-30f ? d * boost : 1e-30f;   // 64 rejections in a row: p < 1e-80
}

LZ_RNG_HD float uniform_draw(uint64_t seed, int64_t game, int64_t ply, uint32_t purpose) {
    return u01(draw(seed, game, ply, purpose, 0u, 0u).x);
}

}  // namespace lzrng
