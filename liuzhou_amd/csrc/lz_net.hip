// lz_net.hip -- fused policy + bucketed-value ResNet forward for 6x6 Liuzhou boards on gfx950.
//
// One persistent workgroup (8 waves, two per SIMD) owns S samples (S*36 board cells) and runs the WHOLE
// network on them without touching HBM in between:
//   * the fp32 residual stream lives in MFMA accumulator registers for the whole trunk
//     (each wave owns 9 tiles of 16 cells x 2 tiles of 16 channels = 72 registers, plus 72 for the
//     conv1 output: 144 accumulators fit the AGPR half of the 256-register budget of 2 waves/SIMD);
//   * conv inputs are staged as fp16 [cell][channel] rows in LDS (one buffer, rewritten per layer);
//   * every 3x3 conv is 9 shifted GEMMs on v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand
//     (pre-packed in fragment order, streamed from L2 with one 16-byte load per lane) and the
//     activations as the B operand (one ds_read_b128 per lane; out-of-board taps read a zero row),
//     so the D tile has the cell on the lane and 4 consecutive channels in registers -> 8-byte LDS
//     writes for the next layer;
//   * BatchNorm is folded at pack time (liuzhou_amd/net_pack.py); heads (global pooling, small FCs,
//     log-softmax, bucket expectation) run on the same workgroup from LDS.
// Reference architecture: src/neural_network.py:67-259 (ChessNet.forward).  fp16 operands, fp32
// accumulate -- the counterpart of the reference's autocast-fp16 inference (v1/python/mcts_gpu.py:640-646).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <utility>
#include <vector>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liuzhou_hip.h"
#include "lz_net_dev.h"

namespace {

template <int C, int S, int W>
__global__ __launch_bounds__(W * 64, (C == 128 && W == 4) ? 1 : 2) void net_forward_kernel(NetParams P, const float* __restrict__ planes,
                                                            const uint64_t* __restrict__ packed,
                                                            int64_t N, float* __restrict__ lp1,
                                                            float* __restrict__ lp2, float* __restrict__ lpm,
                                                            float* __restrict__ vlogits, float* __restrict__ value) {
    if (P.n_dev != nullptr) { const long long nd = *P.n_dev; N = nd < N ? nd : N; }   // count produced on the device
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    NetCtx<C, S, W> ctx;
#ifdef LZ_EXP_HEAD_STAMPS
    ctx.t_entry = __builtin_readcyclecounter();
#endif
    net_setup<C, S, W>(P, lds, ctx);
#ifdef LZ_EXP_HEAD_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ctx.t_setup = __builtin_readcyclecounter();
#endif
    // diagnostic (LZ_NET_DEBUG_STOP=99): shader clock held by this kernel = s_memtime ticks per 100 MHz wall tick
    const uint64_t dbg_t0 = P.debug_stop == 99 ? __builtin_readcyclecounter() : 0;
    const uint64_t dbg_w0 = P.debug_stop == 99 ? wall_clock64() : 0;
    const int64_t n_pass = (N + S - 1) / S;
    for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
        const int64_t n0 = pass * S;
        const int nvalid = (int)((N - n0) < S ? (N - n0) : S);
        net_pass<C, S, W>(P, lds, ctx, planes, packed, n0, nvalid, lp1, lp2, lpm, vlogits, value);
    }
    if (P.debug_stop == 99 && blockIdx.x == 7 && threadIdx.x == 0) {
        const uint64_t dt = __builtin_readcyclecounter() - dbg_t0, dw = wall_clock64() - dbg_w0;
        value[0] = (float)dt; value[1] = (float)dw;
    }
}

template <int C, int S, int W>
int launch_net(const NetParams& P, const float* planes, const uint64_t* packed, int64_t N, float* lp1, float* lp2,
               float* lpm, float* vlogits, float* value, int max_blocks, hipStream_t st) {
    using K = Cfg<C, S, W>;
    auto kern = net_forward_kernel<C, S, W>;
    const int64_t n_pass = (N + S - 1) / S;
    int grid = (int)(n_pass < max_blocks ? n_pass : max_blocks);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(K::THREADS), K::LDS_BYTES, st, P, planes, packed, N, lp1, lp2, lpm, vlogits, value);
    return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}

template <int C, int S, int W>
int configure_net() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(net_forward_kernel<C, S, W>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<C, S, W>::LDS_BYTES) == hipSuccess
               ? LZ_OK : LZ_ERR_LAUNCH;
}

}  // namespace

// ---- optional live timing of the network kernel (bench.py roofline): HIP events on the launch stream ----
namespace {
struct NetProf {
    bool on = false;
    int used = 0;
    static constexpr int kMax = 8192;
    hipEvent_t ev[2 * kMax];
    bool created = false;
    double flops = 0.0;
    int64_t evals = 0;
} g_prof;
}  // namespace

extern "C" void lz_prof_aux_reset(void);

extern "C" {

int lz_prof_enable(int on) {
    if (on && !g_prof.created) {
        for (int i = 0; i < 2 * NetProf::kMax; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return LZ_ERR_LAUNCH;
        g_prof.created = true;
    }
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.evals = 0;
    lz_prof_aux_reset();
    return LZ_OK;
}

/* call after the stream has been synchronised: total / count of the timed network launches */
int lz_prof_net_summary(double* total_ms, int64_t* launches, int64_t* evals) {
    double t = 0.0;
    for (int i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return LZ_ERR_LAUNCH;
        t += ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = g_prof.used;
    if (evals) *evals = g_prof.evals;
    return LZ_OK;
}

/* Launches bracketed on several streams may overlap: `busy_ms` is the length of the union of their [start, end]
 * intervals (event times relative to the first recorded event), i.e. the time during which at least one network
 * kernel was running; equal to total_ms when everything ran on one stream. */
int lz_prof_net_busy(double* busy_ms) {
    if (!busy_ms) return LZ_ERR_ARG;
    *busy_ms = 0.0;
    const int n = g_prof.used;
    if (n == 0) return LZ_OK;
    std::vector<std::pair<float, float>> iv((size_t)n);
    for (int i = 0; i < n; ++i) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, g_prof.ev[0], g_prof.ev[2 * i]) != hipSuccess ||
            hipEventElapsedTime(&b, g_prof.ev[0], g_prof.ev[2 * i + 1]) != hipSuccess)
            return LZ_ERR_LAUNCH;
        iv[(size_t)i] = {a, b};
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0.0;
    float lo = iv[0].first, hi = iv[0].second;
    for (int i = 1; i < n; ++i) {
        if (iv[(size_t)i].first > hi) { busy += (double)(hi - lo); lo = iv[(size_t)i].first; hi = iv[(size_t)i].second; }
        else if (iv[(size_t)i].second > hi) hi = iv[(size_t)i].second;
    }
    *busy_ms = busy + (double)(hi - lo);
    return LZ_OK;
}

/* internal (not exported): bracket any other kernel of the path like a network launch -- the persistent search kernel
 * of lz_search.hip, whose `evals` network evaluations happen inside one launch */
int lz_prof_mark_begin(void* stream) {
    if (!(g_prof.on && g_prof.used < NetProf::kMax)) return LZ_OK;
    return hipEventRecord(g_prof.ev[2 * g_prof.used], reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}
int lz_prof_mark_end(void* stream, int64_t evals) {
    if (!(g_prof.on && g_prof.used < NetProf::kMax)) return LZ_OK;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], reinterpret_cast<hipStream_t>(stream));
    g_prof.used += 1; g_prof.evals += evals;
    return LZ_OK;
}

/* Secondary brackets for the HBM-bound kernels of the search (bench.py `roofline.secondary`): kind 0 = the fused
 * expand + backup + select kernel of a simulation, kind 1 = the subtree compaction of a move.  Same mechanism as the
 * network brackets (events on the launch stream, only while lz_prof_enable(1)); begin / end are internal. */
namespace {
struct AuxProf {
    static constexpr int kKinds = 2, kMax = 4096;
    hipEvent_t ev[kKinds][2 * kMax];
    int used[kKinds] = {0, 0};
    int64_t units[kKinds] = {0, 0};
    bool created = false;
} g_aux;
}  // namespace

void lz_prof_aux_reset(void) {
    for (int k = 0; k < AuxProf::kKinds; ++k) { g_aux.used[k] = 0; g_aux.units[k] = 0; }
}
int lz_prof_aux_begin(int kind, void* stream) {
    if (!g_prof.on || kind < 0 || kind >= AuxProf::kKinds) return LZ_OK;
    if (!g_aux.created) {
        for (int k = 0; k < AuxProf::kKinds; ++k)
            for (int i = 0; i < 2 * AuxProf::kMax; ++i)
                if (hipEventCreate(&g_aux.ev[k][i]) != hipSuccess) return LZ_ERR_LAUNCH;
        g_aux.created = true;
    }
    if (g_aux.used[kind] >= AuxProf::kMax) return LZ_OK;
    return hipEventRecord(g_aux.ev[kind][2 * g_aux.used[kind]], reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}
int lz_prof_aux_end(int kind, void* stream, int64_t units) {
    if (!g_prof.on || !g_aux.created || kind < 0 || kind >= AuxProf::kKinds || g_aux.used[kind] >= AuxProf::kMax) return LZ_OK;
    (void)hipEventRecord(g_aux.ev[kind][2 * g_aux.used[kind] + 1], reinterpret_cast<hipStream_t>(stream));
    g_aux.used[kind] += 1; g_aux.units[kind] += units;
    return LZ_OK;
}
/* exported: call after synchronising; lz_prof_enable() resets the counts */
int lz_prof_aux_summary(int kind, double* total_ms, int64_t* launches, int64_t* units) {
    if (kind < 0 || kind >= AuxProf::kKinds) return LZ_ERR_ARG;
    double t = 0.0;
    for (int i = 0; i < g_aux.used[kind]; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_aux.ev[kind][2 * i], g_aux.ev[kind][2 * i + 1]) != hipSuccess) return LZ_ERR_LAUNCH;
        t += ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = g_aux.used[kind];
    if (units) *units = g_aux.units[kind];
    return LZ_OK;
}

int64_t lz_net_desc_bytes(void) { return (int64_t)sizeof(LzNetDesc); }

int lz_net_configure(void) {
    const int a = configure_net<64, 16, 8>(), b = configure_net<128, 8, 8>(), c = configure_net<64, 8, 4>(),
              e = configure_net<128, 8, 4>();
    return a != LZ_OK ? a : (b != LZ_OK ? b : (c != LZ_OK ? c : e));
}

// fp32-operand parity mode (lz_net_f32.hip)
int lz_net_forward_f32_dispatch(const LzNetDesc* d, const float* planes, const uint64_t* packed, int64_t N, float* lp1,
                                float* lp2, float* lpmc, float* value_logits, float* value, const int64_t* n_dev,
                                void* stream);

static int net_forward_impl(const LzNetDesc* d, const float* planes, const uint64_t* packed, int64_t N, float* lp1,
                            float* lp2, float* lpmc, float* value_logits, float* value, void* stream,
                            const int64_t* n_dev = nullptr) {
    if (!d || N < 0) return LZ_ERR_ARG;
    if (N == 0) return LZ_OK;
    if (!d->wfrag || !d->fparams || (!planes && !packed)) return LZ_ERR_ARG;
    const bool heads = lp1 && lp2 && lpmc;
    if (!heads && (lp1 || lp2 || lpmc || !value)) return LZ_ERR_ARG;     // all three policy outputs, or values only
    if (d->blocks < 0 || d->blocks > (LZ_NET_MAX_LAYERS - 2) / 2 || d->num_layers != 2 + 2 * d->blocks) return LZ_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d->wfrag) & 15) || (reinterpret_cast<uintptr_t>(d->fparams) & 15)) return LZ_ERR_ALIGN;
    if (d->flags & (4 | 8))
        return lz_net_forward_f32_dispatch(d, planes, packed, N, lp1, lp2, lpmc, value_logits, value, n_dev, stream);
    NetParams P = make_net_params(d);
#ifdef LZ_EXP_SAME_LAYER    /* compile-time only, like the other LZ_EXP_* experiments: never in the shipped library.  Timing
                             * experiment (results are wrong): every trunk conv reads the first block's weights, so the weight
                             * set fits the 4 MB XCD L2 -- bounds what L2 misses on the 5.9 MB set of 10x128 cost */
    for (int i = 3; i < 1 + 2 * d->blocks; ++i) P.layer_off[i] = d->layer_offsets[1 + ((i - 1) & 1)];
#endif
    P.n_dev = reinterpret_cast<const long long*>(n_dev);
    P.debug_stop = getenv("LZ_NET_DEBUG_STOP") ? atoi(getenv("LZ_NET_DEBUG_STOP")) : 0;
    // flags bit 0 (64 channels): 4-wave workgroups of 8 samples, two per CU -- twice as many workgroups per batch, so
    // that a half-size batch still covers every CU when two of them are evaluated concurrently on two streams
    const bool half_wg = d->channels == 64 && (d->flags & 1);
    const int max_blocks = d->max_blocks > 0 ? d->max_blocks : (half_wg ? 512 : 256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->channels != 64 && d->channels != 128) return LZ_ERR_UNSUPPORTED;
    const bool prof = g_prof.on && g_prof.used < NetProf::kMax;
    if (prof) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
    // Two 4-wave workgroups of 8 samples per CU (<64,8,4>) were measured, also staggered by half a layer: 9 % fewer
    // shader cycles per pass, but the chip then holds 1.87 GHz instead of 2.05 GHz -- the same wall time.
    // flags bit 1 (128 channels): 4-wave workgroups, one wave per SIMD with 512 registers and 4 channel tiles per wave
    // (half the LDS operand reads of the 8-wave shape)
    const bool wide = d->channels == 128 && (d->flags & 2);
    const int rc = d->channels == 64
                       ? (half_wg
                              ? launch_net<64, 8, 4>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st)
                              : launch_net<64, 16, 8>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st))
                       : (wide ? launch_net<128, 8, 4>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st)
                               : launch_net<128, 8, 8>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st));
    if (prof) { (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st); g_prof.used += 1; g_prof.evals += N; }
    return rc;
}

int lz_net_forward_f16(const LzNetDesc* d, const float* planes, int64_t N, float* lp1, float* lp2, float* lpmc,
                       float* value_logits, float* value, void* stream) {
    return net_forward_impl(d, planes, nullptr, N, lp1, lp2, lpmc, value_logits, value, stream);
}

int lz_net_forward_packed_f16(const LzNetDesc* d, const void* packed_states, int64_t N, float* lp1, float* lp2,
                              float* lpmc, float* value_logits, float* value, void* stream) {
    return net_forward_impl(d, nullptr, reinterpret_cast<const uint64_t*>(packed_states), N, lp1, lp2, lpmc,
                            value_logits, value, stream);
}

int lz_net_forward_packed_counted_f16(const LzNetDesc* d, const void* packed_states, int64_t capacity,
                                      const int64_t* count, float* lp1, float* lp2, float* lpmc, float* value_logits,
                                      float* value, void* stream) {
    if (!count) return LZ_ERR_ARG;
    return net_forward_impl(d, nullptr, reinterpret_cast<const uint64_t*>(packed_states), capacity, lp1, lp2, lpmc,
                            value_logits, value, stream, count);
}

}  // extern "C"
