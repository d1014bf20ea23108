// lz_net_f32.hip -- the same fused policy + bucketed-value ResNet forward with fp32 OPERANDS: the parity mode of the
// network kernel (LzNetDesc.flags bit 2).  Where lz_net.hip is the reference's autocast-fp16 inference
// (v1/python/mcts_gpu.py:640-646) tuned to the MFMA roofline, this kernel is the reference's fp32 forward
// (src/neural_network.py:213-259 in eval mode) and exists to meet its <= 1e-5 tolerance on policy / value tensors with a
// hand-written kernel; it is written for clarity, not speed (v_mfma_f32_16x16x4_f32 is a 1/16-rate instruction next to
// the fp16 one).
//
// One workgroup (4 waves) owns S samples for the whole network, activations never leave the CU:
//   * conv input: fp32 rows [cell][channel] on the zero-bordered 7-wide board of lz_net.hip (58 rows per sample, so a
//     3x3 tap is a constant row offset and borders read zeros);
//   * every conv is taps x K/4 MFMAs v_mfma_f32_16x16x4_f32 per (16-cell, 16-channel) tile, weights as the A operand
//     (fp32 fragments [layer][tap][K/4][Cout/16][64 lanes], BatchNorm folded at pack time), activations as B;
//     D has the cell on the lane and 4 consecutive channels in registers -> float4 LDS stores for the next layer;
//   * the fp32 residual stream lives in accumulator registers; heads (global pooling, dense layers, log-softmax,
//     bucket expectation) are plain fp32 loops over LDS.
//
// SPLIT-OPERAND MODE (round 6, LzNetDesc.flags bit 3, template flag X3): the same kernel with every conv operand held as
// TWO fp16 numbers, v = hi + lo * 2^-11 (hi = fp16(v), lo = fp16((v - hi) * 2^11): 22 significant bits, the low half
// scaled so that it stays a normal fp16 number), and every product as THREE v_mfma_f32_16x16x32_f16:
//   W * a  ~=  Wh*ah  +  2^-11 * (Wh*al + Wl*ah)          (the dropped Wl*al term is 2^-22 relative)
// fp16 x fp16 products are exact in the fp32 accumulators, so the result is the fp32 convolution of operands rounded to 22
// bits: <= 1e-5 on every network output like the fp32-operand mode, at 3/16 of its matrix-pipe time.  The conv input rows
// hold [hi(C) | lo(C)] halfs (the bytes of one fp32 row, + 16 bytes of padding against bank conflicts); the weights' low
// halves come from net_pack (`wfrag_lo`, same fragment order and offsets as `wfrag`); two accumulator sets (the main
// one and the 2^11-scaled cross terms) are merged after every conv.  Heads are the fp32 loops of the fp32 mode.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liuzhou_hip.h"
#include "lz_wave.h"

namespace lzf32 {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr float kLoUp = 2048.0f, kLoDown = 1.0f / 2048.0f;      // 2^11: scale of the low halves (net_pack.LO_SHIFT)

constexpr int kHead = 64, kMlp = 128, kBins = 101, kPool = 3 * kHead;
constexpr int kThreads = 256, kWaves = 4, kWave = 64;

struct Params {
    const _Float16* wh;         // split-operand mode: fp16 conv fragments (LzNetDesc.wfrag) ...
    const _Float16* wl;         // ... and their low halves (LzNetDesc.wfrag_lo)
    const float* w;             // fp32 conv fragments
    const float* fp;            // per-channel parameters + dense head matrices (the fp16 kernel's fparams)
    int layer_off[LZ_NET_MAX_LAYERS];   // element offsets of stem, (conv1, conv2) x blocks, stacked head convs
    int blocks;
    int stem_bias, blk0, trunk_a, trunk_b, head_bias, p_gwT, p_a2, p_b2, p_out, v_w1T, v_b1, v_w2T, v_b2;
    const long long* n_dev;
};

template <int C, bool X3 = false>
struct Cfg {
    static constexpr int S = C == 128 ? 2 : 4;                  // samples per workgroup
    static constexpr int NPOS = S * 36;
    static constexpr int NT = (NPOS + 15) / 16;                 // 16-cell tiles (the last one may be partly padding)
    static constexpr int CT = C / 16;
    static constexpr int CTW = CT / kWaves;                     // channel tiles per wave in the trunk (1 or 2)
    static constexpr int ROWS = S * 58 + 1;                     // + one all-zero row for padding cells
    static constexpr int ZROW = S * 58;
    // split mode: a row is [hi(C) | lo(C)] halfs + 8 halfs of padding (an odd number of 16-byte slots per row, so that the
    // 16 cells of a tile do not meet in one bank group); in floats: C + 4
    static constexpr int ROW_HALFS = 2 * C + 8;
    static constexpr int ACT_FLOATS = X3 ? ROWS * (C + 4) : ROWS * C;
    static constexpr int MAP_FLOATS = NPOS * kHead;             // one 64-channel head map [cell][channel]
    static constexpr int POOL_OFF = ACT_FLOATS + 2 * MAP_FLOATS;
    static constexpr int G_OFF = POOL_OFF + S * kPool;
    static constexpr int HID_OFF = G_OFF + S * kHead;
    static constexpr int VLOG_OFF = HID_OFF + S * kMlp;
    static constexpr int PLOG_OFF = VLOG_OFF + S * 112;
    static constexpr int LDS_FLOATS = PLOG_OFF + S * 3 * 36;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

__host__ __device__ constexpr int board_row(int n) {            // row of board cell n = 36*sample + 6*r + c
    const int s = n / 36, p = n - s * 36;
    const int r = p / 6, c = p - r * 6;
    return s * 58 + 1 + 7 * (r + 1) + c;
}

// acc[i][j] += W(layer)[channel tiles ct0 + j] * act for all taps / K blocks.  `row` = this lane's cell row per tile.
template <int C, int NT, int NW, int TAPS, int KDIM, int CTN>
__device__ __forceinline__ void conv(f4 (&acc)[NT][NW], const float* __restrict__ w, int layer_off, int ct0,
                                     const float* act, const int (&row)[NT], int lane) {
    const int k = lane >> 4;
    for (int t = 0; t < TAPS; ++t) {
        const int toff = TAPS == 9 ? ((t / 3) - 1) * 7 + (t % 3) - 1 : 0;
        for (int kb = 0; kb < KDIM / 4; ++kb) {
            float a[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) a[j] = w[layer_off + ((t * (KDIM / 4) + kb) * CTN + ct0 + j) * 64 + lane];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float b = act[(row[i] + toff) * C + kb * 4 + k];
#pragma unroll
                for (int j = 0; j < NW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc[i][j], 0, 0, 0);
            }
        }
    }
}

// split-operand conv: main[i][j] += Wh*ah, cross[i][j] += Wh*al + Wl*ah (both low halves carry a factor 2^11); K blocks of 32
// channels, weights = A operand of v_mfma_f32_16x16x32_f16 in the fp16 kernel's fragment order.  The weight fragments are
// fetched TWO K steps ahead into three register sets used in rotation (the step loop is unrolled by three, so the rotation
// is a renaming, not a copy that would wait for the load): with one wave per SIMD a K step is ~450 matrix cycles, an L2
// round trip more than that -- one step ahead left half of every load's latency on the critical path.  Measured
// (profiles/r06_experiments.md section 5): 10x128 0.80 -> 1.26 M evaluations/s; 6x64 6.20 -> 5.96 M, so the 64-channel
// kernel (9 cell tiles per fragment: a K step is already longer than the load) keeps ONE step ahead (`AHEAD`).
template <int C, int NT, int NW, int TAPS, int KDIM, int CTN>
__device__ __forceinline__ void conv_x3(f4 (&main)[NT][NW], f4 (&cross)[NT][NW], const _Float16* __restrict__ wh,
                                        const _Float16* __restrict__ wl, int layer_off, int ct0, const _Float16* act,
                                        const int (&row)[NT], int lane) {
    constexpr int RS = 2 * C + 8, KB = KDIM / 32, STEPS = TAPS * KB;
    const int k4 = lane >> 4;
    const _Float16* whl = wh + layer_off + (size_t)ct0 * 512 + lane * 8;
    const _Float16* wll = wl + layer_off + (size_t)ct0 * 512 + lane * 8;
    struct Frag { h8 h[NW], l[NW]; };
    auto fetch = [&](Frag& f, int s) {
        const int sn = s < STEPS ? s : STEPS - 1;               // past the end: re-read the last step's (unused)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            f.h[j] = *reinterpret_cast<const h8*>(whl + ((size_t)sn * CTN + j) * 512);
            f.l[j] = *reinterpret_cast<const h8*>(wll + ((size_t)sn * CTN + j) * 512);
        }
    };
    auto step = [&](const Frag& a, int s) {
        const int t = s / KB, kb = s - t * KB;
        const int toff = TAPS == 9 ? ((t / 3) - 1) * 7 + (t % 3) - 1 : 0;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const _Float16* r = act + (row[i] + toff) * RS + kb * 32 + k4 * 8;
            const h8 bh = *reinterpret_cast<const h8*>(r);
            const h8 bl = *reinterpret_cast<const h8*>(r + C);
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                main[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[j], bh, main[i][j], 0, 0, 0);
                cross[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[j], bl, cross[i][j], 0, 0, 0);
                cross[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l[j], bh, cross[i][j], 0, 0, 0);
            }
        }
    };
    constexpr int AHEAD = C == 128 ? 2 : 1;
    if constexpr (AHEAD == 2) {
        Frag a0, a1, a2;
        fetch(a0, 0);
        fetch(a1, 1);
        int s = 0;
        for (; s + 3 <= STEPS; s += 3) {
            fetch(a2, s + 2); step(a0, s);
            fetch(a0, s + 3); step(a1, s + 1);
            fetch(a1, s + 4); step(a2, s + 2);
        }
        if constexpr (STEPS % 3 >= 1) step(a0, s);
        if constexpr (STEPS % 3 == 2) step(a1, s + 1);
    } else {
        Frag a0, a1;
        fetch(a0, 0);
        int s = 0;
        for (; s + 2 <= STEPS; s += 2) {
            fetch(a1, s + 1); step(a0, s);
            fetch(a0, s + 2); step(a1, s + 1);
        }
        if constexpr (STEPS % 2 == 1) step(a0, s);
    }
}

// global pooling of a [cell][64] map -> pooled[s][192] = mean | max | sqrt(var + 1e-6)  (src/neural_network.py:67-80,
// var with unbiased=False, two passes)
template <int S>
__device__ __forceinline__ void gpool(const float* map, float* pooled, int tid) {
    for (int o = tid; o < S * kHead; o += kThreads) {
        const int s = o / kHead, c = o - s * kHead;
        float sum = 0.f, mx = -INFINITY;
        for (int p = 0; p < 36; ++p) { const float v = map[(s * 36 + p) * kHead + c]; sum += v; mx = fmaxf(mx, v); }
        const float mean = sum / 36.0f;
        float sq = 0.f;
        for (int p = 0; p < 36; ++p) { const float d = map[(s * 36 + p) * kHead + c] - mean; sq += d * d; }
        pooled[s * kPool + c] = mean;
        pooled[s * kPool + kHead + c] = mx;
        pooled[s * kPool + 2 * kHead + c] = sqrtf(sq / 36.0f + 1e-6f);
    }
}

template <int C, bool X3 = false>
__global__ __launch_bounds__(kThreads) void net_forward_f32_kernel(Params P, const float* __restrict__ planes,
                                                                   const uint64_t* __restrict__ packed, int64_t N,
                                                                   float* __restrict__ lp1, float* __restrict__ lp2,
                                                                   float* __restrict__ lpm, float* __restrict__ vlogits,
                                                                   float* __restrict__ value) {
    using K = Cfg<C, X3>;
    constexpr int S = K::S, NT = K::NT, NW = K::CTW;
    if (P.n_dev != nullptr) { const long long nd = *P.n_dev; N = nd < N ? nd : N; }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* act = lds;
    _Float16* acth = reinterpret_cast<_Float16*>(lds);        // split mode: rows of [hi(C) | lo(C) | pad] halfs
    constexpr int RS = K::ROW_HALFS;
    float* pmap = lds + K::ACT_FLOATS;
    float* vmap = pmap + K::MAP_FLOATS;
    float* pooled = lds + K::POOL_OFF;
    float* gvec = lds + K::G_OFF;
    float* hid = lds + K::HID_OFF;
    float* vlog = lds + K::VLOG_OFF;
    float* plog = lds + K::PLOG_OFF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* fp = P.fp;
    int row[NT];                                              // this lane's cell row in each tile (zero row: padding)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = i * 16 + (lane & 15);
        row[i] = n < K::NPOS ? board_row(n) : K::ZROW;
    }
    const int ct0 = wave * NW;                                // first trunk channel tile of this wave
    const int chq = (lane >> 4) * 4;                          // first of the lane's 4 consecutive channels in a tile
    const int64_t n_pass = (N + S - 1) / S;
    for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
        const int64_t n0 = pass * S;
        const int nvalid = (int)((N - n0) < S ? (N - n0) : S);
        __syncthreads();
        for (int i = tid; i < K::ACT_FLOATS; i += kThreads) act[i] = 0.f;
        __syncthreads();
        // ---- input planes (src/neural_network.py:15-65): channels 0..10 of the 32-wide stem K block ----
        for (int o = tid; o < K::NPOS * 11; o += kThreads) {
            const int n = o / 11, ch = o - n * 11;
            const int s = n / 36, p = n - s * 36;
            float v = 0.f;
            if (s < nvalid) {
                if (packed != nullptr) {
                    const uint64_t* rec = packed + (n0 + s) * 4;
                    const uint64_t w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3];
                    const bool white = (w0 >> 53) & 1;
                    const int phase = (int)((w0 >> 50) & 7);
                    const uint64_t src = ch == 0 ? (white ? w1 : w0) : ch == 1 ? (white ? w0 : w1)
                                       : ch == 2 ? (white ? w3 : w2) : ch == 3 ? (white ? w2 : w3) : 0ull;
                    v = ch < 4 ? (float)((src >> p) & 1) : (phase == ch - 3 ? 1.f : 0.f);
                } else {
                    v = planes[(n0 + s) * 396 + ch * 36 + p];
                }
            }
            if constexpr (X3) acth[board_row(n) * RS + ch] = (_Float16)v;      // planes are 0 / 1: exact, low half 0
            else act[board_row(n) * C + ch] = v;
        }
        __syncthreads();
        f4 x[NT][NW], acc[NT][NW];
        f4 cr[NT][NW];                                        // split mode: the 2^11-scaled cross terms of the running conv
        auto zero_cr = [&]() {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) cr[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        };
        auto merge_cr = [&](f4 (&into)[NT][NW]) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) into[i][j] = into[i][j] + cr[i][j] * kLoDown;
        };
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j) x[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        // ---- stem: x = relu(conv(planes) + bias) ----
        if constexpr (X3) {
            zero_cr();
            conv_x3<C, NT, NW, 9, 32, K::CT>(x, cr, P.wh, P.wl, P.layer_off[0], ct0, acth, row, lane);
            merge_cr(x);
        } else {
            conv<C, NT, NW, 9, 32, K::CT>(x, P.w, P.layer_off[0], ct0, act, row, lane);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const f4 b = *reinterpret_cast<const f4*>(fp + P.stem_bias + (ct0 + j) * 16 + chq);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const f4 v = x[i][j] + b;
                x[i][j] = (f4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            }
        }
        // store relu(sc * v + sh) (or relu(v + sh)) of the lane's channels as the next conv's input rows
        auto store = [&](const f4 (&v)[NT][NW], int sc_off, int sh_off) {
            __syncthreads();                                  // everyone finished reading the previous input
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int ch = (ct0 + j) * 16 + chq;
                const f4 sh = *reinterpret_cast<const f4*>(fp + sh_off + ch);
                f4 sc = (f4){1.f, 1.f, 1.f, 1.f};
                if (sc_off >= 0) sc = *reinterpret_cast<const f4*>(fp + sc_off + ch);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (row[i] == K::ZROW) continue;          // padding cells of the last tile
                    const f4 t = v[i][j] * sc + sh;
                    const f4 r = (f4){fmaxf(t[0], 0.f), fmaxf(t[1], 0.f), fmaxf(t[2], 0.f), fmaxf(t[3], 0.f)};
                    if constexpr (X3) {
                        const h4 hi = __builtin_convertvector(r, h4);
                        const h4 lo = __builtin_convertvector((r - __builtin_convertvector(hi, f4)) * kLoUp, h4);
                        *reinterpret_cast<h4*>(acth + row[i] * RS + ch) = hi;
                        *reinterpret_cast<h4*>(acth + row[i] * RS + C + ch) = lo;
                    } else {
                        *reinterpret_cast<f4*>(act + row[i] * C + ch) = r;
                    }
                }
            }
            __syncthreads();
        };
        // ---- pre-activation residual blocks (src/neural_network.py:82-95): x += conv2(relu(bn2(conv1(relu(bn1(x)))))) ----
        for (int blk = 0; blk < P.blocks; ++blk) {
            const int bp = P.blk0 + blk * 3 * C;              // a1 | b1 | bias1 (bn2 folded into conv1)
            store(x, bp, bp + C);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
            if constexpr (X3) {
                zero_cr();
                conv_x3<C, NT, NW, 9, C, K::CT>(acc, cr, P.wh, P.wl, P.layer_off[1 + 2 * blk], ct0, acth, row, lane);
                merge_cr(acc);
            } else {
                conv<C, NT, NW, 9, C, K::CT>(acc, P.w, P.layer_off[1 + 2 * blk], ct0, act, row, lane);
            }
            store(acc, -1, bp + 2 * C);
            if constexpr (X3) {
                zero_cr();
                conv_x3<C, NT, NW, 9, C, K::CT>(x, cr, P.wh, P.wl, P.layer_off[2 + 2 * blk], ct0, acth, row, lane);
                merge_cr(x);
            } else {
                conv<C, NT, NW, 9, C, K::CT>(x, P.w, P.layer_off[2 + 2 * blk], ct0, act, row, lane);
            }
        }
        // ---- trunk output relu(bn(x)); head 1x1 convs: 8 output tiles (policy 0..3 | value 4..7), 2 per wave ----
        store(x, P.trunk_a, P.trunk_b);
        f4 h[NT][2];
#pragma unroll
        for (int i = 0; i < NT; ++i) { h[i][0] = (f4){0.f, 0.f, 0.f, 0.f}; h[i][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
        if constexpr (X3) {
            f4 hc[NT][2];
#pragma unroll
            for (int i = 0; i < NT; ++i) { hc[i][0] = (f4){0.f, 0.f, 0.f, 0.f}; hc[i][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
            conv_x3<C, NT, 2, 1, C, 8>(h, hc, P.wh, P.wl, P.layer_off[1 + 2 * P.blocks], wave * 2, acth, row, lane);
#pragma unroll
            for (int i = 0; i < NT; ++i) { h[i][0] = h[i][0] + hc[i][0] * kLoDown; h[i][1] = h[i][1] + hc[i][1] * kLoDown; }
        } else {
            conv<C, NT, 2, 1, C, 8>(h, P.w, P.layer_off[1 + 2 * P.blocks], wave * 2, act, row, lane);
        }
        {
            float* map = wave < 2 ? pmap : vmap;              // waves 0,1 -> policy map, 2,3 -> value map
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hc = (wave * 2 + j) * 16 + chq;     // channel among the 128 stacked head channels
                const f4 b = *reinterpret_cast<const f4*>(fp + P.head_bias + hc);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int n = i * 16 + (lane & 15);
                    if (n >= K::NPOS) continue;
                    const f4 t = h[i][j] + b;
                    *reinterpret_cast<f4*>(map + n * kHead + (hc & 63)) =
                        (f4){fmaxf(t[0], 0.f), fmaxf(t[1], 0.f), fmaxf(t[2], 0.f), fmaxf(t[3], 0.f)};
                }
            }
        }
        __syncthreads();
        // ---- policy head (src/neural_network.py:97-124) ----
        if (lp1 != nullptr) {
            gpool<S>(pmap, pooled, tid);
            __syncthreads();
            for (int o = tid; o < S * kHead; o += kThreads) {         // g = gpool_linear(pooled), no bias
                const int s = o / kHead, c = o - s * kHead;
                float a = 0.f;
                for (int q = 0; q < kPool; ++q) a += pooled[s * kPool + q] * fp[P.p_gwT + q * kHead + c];
                gvec[o] = a;
            }
            __syncthreads();
            for (int o = tid; o < K::NPOS * 3; o += kThreads) {       // three 1x1 output convs on relu(bn2(p + g))
                const int n = o / 3, hd = o - n * 3, s = n / 36;
                float a = 0.f;
                for (int c = 0; c < kHead; ++c) {
                    const float p2 = fmaxf((pmap[n * kHead + c] + gvec[s * kHead + c]) * fp[P.p_a2 + c] + fp[P.p_b2 + c], 0.f);
                    a += p2 * fp[P.p_out + hd * kHead + c];
                }
                plog[(s * 3 + hd) * 36 + (n - s * 36)] = a;
            }
            __syncthreads();
            for (int r = wave; r < S * 3; r += kWaves) {               // log_softmax over the 36 cells, one wave per row
                const int s = r / 3, hd = r - s * 3;
                const float v = lane < 36 ? plog[r * 36 + lane] : -INFINITY;
                const float mx = lzw::wave_max(v);
                const float e = lzw::wave_sum(lane < 36 ? expf(v - mx) : 0.f);
                const float lse = mx + logf(e);
                if (lane < 36 && s < nvalid) (hd == 0 ? lp1 : hd == 1 ? lp2 : lpm)[(n0 + s) * 36 + lane] = v - lse;
            }
        }
        // ---- value head (src/neural_network.py:126-148) + bucket expectation (:201-210) ----
        gpool<S>(vmap, pooled, tid);
        __syncthreads();
        for (int o = tid; o < S * kMlp; o += kThreads) {
            const int s = o / kMlp, c = o - s * kMlp;
            float a = fp[P.v_b1 + c];
            for (int q = 0; q < kPool; ++q) a += pooled[s * kPool + q] * fp[P.v_w1T + q * kMlp + c];
            hid[o] = fmaxf(a, 0.f);
        }
        __syncthreads();
        for (int o = tid; o < S * kBins; o += kThreads) {
            const int s = o / kBins, c = o - s * kBins;
            float a = fp[P.v_b2 + c];
            for (int q = 0; q < kMlp; ++q) a += hid[s * kMlp + q] * fp[P.v_w2T + q * kBins + c];
            vlog[s * 112 + c] = a;
        }
        __syncthreads();
        for (int s = wave; s < nvalid; s += kWaves) {
            const float v0 = vlog[s * 112 + lane];
            const float v1 = lane + 64 < kBins ? vlog[s * 112 + lane + 64] : -INFINITY;
            const float mx = lzw::wave_max(fmaxf(v0, v1));
            const float e0 = expf(v0 - mx), e1 = lane + 64 < kBins ? expf(v1 - mx) : 0.f;
            const float sum = lzw::wave_sum(e0 + e1);
            const float ex = lzw::wave_sum(e0 / sum * (-1.0f + 0.02f * (float)lane) +
                                           e1 / sum * (-1.0f + 0.02f * (float)(lane + 64)));
            if (lane == 0 && value != nullptr) value[n0 + s] = ex;
            if (vlogits != nullptr) {
                vlogits[(n0 + s) * kBins + lane] = v0;
                if (lane + 64 < kBins) vlogits[(n0 + s) * kBins + lane + 64] = v1;
            }
        }
    }
}

template <int C, bool X3 = false>
int launch(const Params& P, const float* planes, const uint64_t* packed, int64_t N, float* lp1, float* lp2, float* lpm,
           float* vlogits, float* value, int max_blocks, hipStream_t st) {
    using K = Cfg<C, X3>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(net_forward_f32_kernel<C, X3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES) != hipSuccess)
            return LZ_ERR_LAUNCH;
        configured = true;
    }
    const int64_t n_pass = (N + K::S - 1) / K::S;
    int grid = (int)(n_pass < max_blocks ? n_pass : max_blocks);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((net_forward_f32_kernel<C, X3>), dim3(grid), dim3(kThreads), K::LDS_BYTES, st, P, planes, packed, N, lp1,
                       lp2, lpm, vlogits, value);
    return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}

}  // namespace lzf32

// called by net_forward_impl (lz_net.hip) when LzNetDesc.flags bit 2 is set
extern "C" int lz_net_forward_f32_dispatch(const LzNetDesc* d, const float* planes, const uint64_t* packed, int64_t N, float* lp1,
                                float* lp2, float* lpmc, float* value_logits, float* value, const int64_t* n_dev,
                                void* stream) {
    const bool x3 = (d->flags & 8) != 0;                       // split-operand mode: fp16 fragments + their low halves
    if (x3) {
        if (!d->wfrag_lo || (reinterpret_cast<uintptr_t>(d->wfrag_lo) & 15) || (reinterpret_cast<uintptr_t>(d->wfrag) & 15) ||
            d->wfrag_lo_bytes < (int64_t)d->layer_offsets[d->num_layers - 1] * 2)
            return LZ_ERR_ARG;
    } else if (!d->wfrag_f32 || (reinterpret_cast<uintptr_t>(d->wfrag_f32) & 15)) {
        return LZ_ERR_ARG;
    }
    lzf32::Params P;
    P.w = d->wfrag_f32;
    P.wh = reinterpret_cast<const _Float16*>(d->wfrag);
    P.wl = reinterpret_cast<const _Float16*>(d->wfrag_lo);
    P.fp = d->fparams;
    for (int i = 0; i < d->num_layers; ++i) P.layer_off[i] = d->layer_offsets[i];
    P.blocks = d->blocks;
    P.n_dev = reinterpret_cast<const long long*>(n_dev);
    P.stem_bias = d->off_stem_bias; P.blk0 = d->off_block0; P.trunk_a = d->off_trunk_a; P.trunk_b = d->off_trunk_b;
    P.head_bias = d->off_head_bias; P.p_gwT = d->off_p_gwT; P.p_a2 = d->off_p_a2; P.p_b2 = d->off_p_b2;
    P.p_out = d->off_p_out; P.v_w1T = d->off_v_w1T; P.v_b1 = d->off_v_b1; P.v_w2T = d->off_v_w2T; P.v_b2 = d->off_v_b2;
    const int max_blocks = d->max_blocks > 0 ? d->max_blocks : 1024;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (x3) {
        if (d->channels == 64) return lzf32::launch<64, true>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st);
        if (d->channels == 128) return lzf32::launch<128, true>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st);
        return LZ_ERR_UNSUPPORTED;
    }
    if (d->channels == 64) return lzf32::launch<64>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st);
    if (d->channels == 128) return lzf32::launch<128>(P, planes, packed, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st);
    return LZ_ERR_UNSUPPORTED;
}
