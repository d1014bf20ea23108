// lz_net.hip -- fused policy + bucketed-value ResNet forward for 6x6 Liuzhou boards on gfx950.
//
// One persistent workgroup (4 waves, one per SIMD) owns S samples (S*36 board cells) and runs the WHOLE
// network on them without touching HBM in between:
//   * the fp32 residual stream lives in MFMA accumulator registers for the whole trunk
//     (each wave owns 9 tiles of 16 cells x 4 tiles of 16 channels = 144 VGPRs);
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
#include <math.h>
#include <stdint.h>

#include "../../include/liuzhou_hip.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kHead = 64;      // policy / value head channels
constexpr int kMlp = 128;
constexpr int kBins = 101;
constexpr int kPool = 3 * kHead;

struct NetParams {
    const _Float16* wfrag;
    const float* fp;
    int layer_off[32];          // offsets in halfs: stem, (conv1, conv2) x blocks, heads
    int blocks;
    // float-parameter offsets
    int stem_bias, blk0, trunk_a, trunk_b, head_bias, p_gwT, p_a2, p_b2, p_out, v_w1T, v_b1, v_w2T, v_b2;
};

template <int C, int S>
struct Cfg {
    static constexpr int NPOS = S * 36;
    static constexpr int NT = NPOS / 16;
    static constexpr int PG = NT / 9;               // cell groups (4 or 2)
    static constexpr int CG = 4 / PG;               // channel groups (1 or 2)
    static constexpr int CT = C / 16;               // 16-channel output tiles
    static constexpr int KB = C / 32;               // 32-channel K blocks
    static constexpr int KBLOG = (KB == 1) ? 0 : (KB == 2) ? 1 : 2;
    static constexpr int STRIDE = C * 2 + 16;       // bytes per cell row (16-byte aligned, bank-skewed)
    static constexpr int ACT_OFF = 0;
    static constexpr int ZERO_OFF = NPOS * STRIDE;
    static constexpr int POOL_OFF = ZERO_OFF + STRIDE;
    static constexpr int G_OFF = POOL_OFF + S * kPool * 4;
    static constexpr int HID_OFF = G_OFF + S * kHead * 4;
    static constexpr int PLOG_OFF = HID_OFF + S * kMlp * 4;
    static constexpr int PAR_OFF = PLOG_OFF + S * 432;
    static constexpr int LDS_BYTES = PAR_OFF + 5 * kHead * 4;
    static_assert(CT / CG == 4, "each wave owns 4 output-channel tiles");
    static_assert(NT % 9 == 0 && PG * CG == 4, "4 waves per workgroup");
};

// ---- the GEMM core: acc[9 cell tiles][4 channel tiles] += W(layer) * act --------------------------------
// Register budget (one wave per SIMD, 512 registers): 288 accumulators (residual stream + conv1 output),
// weight fragments double-buffered across K steps (2 x 4 x 4), activation fragments in a 2-deep ring
// that runs one cell tile ahead of the MFMAs.
template <int C, int S, bool TAPS9, bool STEM>
__device__ __forceinline__ void step_geometry(int step, int& tap, int& off, int& zoff) {
    using K = Cfg<C, S>;
    int kb;
    if (!TAPS9) { tap = 4; kb = step; }
    else if (STEM) { tap = step; kb = 0; }
    else { tap = step >> K::KBLOG; kb = step & (K::KB - 1); }
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    off = (dy * 6 + dx) * K::STRIDE + kb * 64;
    zoff = kb * 64;
}

template <int C, int S, bool TAPS9, bool STEM>
__device__ __forceinline__ void gemm_step(f4 (&acc)[9][4], const h8 (&A)[4], int step, const unsigned char* lds,
                                          const int (&base)[9], const int (&valid)[9], int zero_addr) {
    int tap, off, zoff;
    step_geometry<C, S, TAPS9, STEM>(step, tap, off, zoff);
    const int za = zero_addr + zoff;
    h8 b0 = *reinterpret_cast<const h8*>(lds + (((valid[0] >> tap) & 1) ? (base[0] + off) : za));
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        h8 b1 = b0;
        if (i + 1 < 9) b1 = *reinterpret_cast<const h8*>(lds + (((valid[i + 1] >> tap) & 1) ? (base[i + 1] + off) : za));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j], b0, acc[i][j], 0, 0, 0);
        b0 = b1;
    }
}

template <int C, int S, bool TAPS9, bool STEM>
__device__ __forceinline__ void conv_gemm(f4 (&acc)[9][4], const h8* __restrict__ wl, int ctn, int ct0,
                                          const unsigned char* lds, const int (&base)[9], const int (&valid)[9],
                                          int zero_addr, int lane) {
    using K = Cfg<C, S>;
    constexpr int nsteps = TAPS9 ? (STEM ? 9 : 9 * K::KB) : K::KB;
    const h8* wp = wl + (size_t)ct0 * 64 + lane;
    const int wstride = ctn * 64;                        // h8 elements per K step
    h8 A0[4], A1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) A0[j] = wp[j * 64];
#pragma unroll 1
    for (int step = 0; step + 1 < nsteps; step += 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) A1[j] = wp[(size_t)(step + 1) * wstride + j * 64];
        gemm_step<C, S, TAPS9, STEM>(acc, A0, step, lds, base, valid, zero_addr);
        if (step + 2 < nsteps) {
#pragma unroll
            for (int j = 0; j < 4; ++j) A0[j] = wp[(size_t)(step + 2) * wstride + j * 64];
        }
        gemm_step<C, S, TAPS9, STEM>(acc, A1, step + 1, lds, base, valid, zero_addr);
    }
    if (nsteps & 1) gemm_step<C, S, TAPS9, STEM>(acc, A0, nsteps - 1, lds, base, valid, zero_addr);
}

__device__ __forceinline__ h4 to_h4(float a, float b, float c, float d) {
    h4 r;
    r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
    return r;
}

// write relu(scale*acc + shift) (per channel) as fp16 rows; `chan_base` = first channel of co tile 0
template <int C, int S, bool HAS_SCALE>
__device__ __forceinline__ void store_act(const f4 (&acc)[9][4], unsigned char* lds, int tile0, int chan_base,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int lane) {
    using K = Cfg<C, S>;
    const int sub = (lane >> 4) * 4;
    f4 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = chan_base + j * 16 + sub;
        sh[j] = *reinterpret_cast<const f4*>(shift + ch);
        if (HAS_SCALE) sc[j] = *reinterpret_cast<const f4*>(scale + ch);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int n = (tile0 + i) * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f4 v = acc[i][j];
            if (HAS_SCALE) v = v * sc[j] + sh[j]; else v = v + sh[j];
            const h4 o = to_h4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
            *reinterpret_cast<h4*>(lds + K::ACT_OFF + n * K::STRIDE + (chan_base + j * 16 + sub) * 2) = o;
        }
    }
}

// global pooling of a [cell][64] fp16 map in LDS -> pooled[s][192] = mean | max | sqrt(var + 1e-6)
template <int C, int S>
__device__ __forceinline__ void gpool64(const unsigned char* lds, float* pooled, int tid) {
    using K = Cfg<C, S>;
    if (tid < S * 16) {
        const int s = tid >> 4, cq = tid & 15;
        float sum[4] = {0.f, 0.f, 0.f, 0.f}, mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int p = 0; p < 36; ++p) {
            const h4 v = *reinterpret_cast<const h4*>(lds + K::ACT_OFF + (s * 36 + p) * K::STRIDE + cq * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float f = (float)v[k]; sum[k] += f; mx[k] = fmaxf(mx[k], f); }
        }
        float mean[4], var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) mean[k] = sum[k] * (1.0f / 36.0f);
        for (int p = 0; p < 36; ++p) {
            const h4 v = *reinterpret_cast<const h4*>(lds + K::ACT_OFF + (s * 36 + p) * K::STRIDE + cq * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = (float)v[k] - mean[k]; var[k] += d * d; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pooled[s * kPool + cq * 4 + k] = mean[k];
            pooled[s * kPool + kHead + cq * 4 + k] = mx[k];
            pooled[s * kPool + 2 * kHead + cq * 4 + k] = sqrtf(var[k] * (1.0f / 36.0f) + 1e-6f);
        }
    }
}

template <int C, int S>
__global__ __launch_bounds__(256, 1) void net_forward_kernel(NetParams P, const float* __restrict__ planes,
                                                            int64_t N, float* __restrict__ lp1,
                                                            float* __restrict__ lp2, float* __restrict__ lpm,
                                                            float* __restrict__ vlogits, float* __restrict__ value) {
    using K = Cfg<C, S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave % K::PG, cg = wave / K::PG;
    const int tile0 = pg * 9;
    const int ct0 = cg * 4;                     // first output-channel tile of this wave
    const int chan0 = ct0 * 16;
    const float* fp = P.fp;

    // per-lane cell geometry of the 9 tiles
    int base[9], valid[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int n = (tile0 + i) * 16 + (lane & 15);
        const int p = n % 36;
        const int r = p / 6, c = p - r * 6;
        base[i] = K::ACT_OFF + n * K::STRIDE + (lane >> 4) * 16;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int rr = r + t / 3 - 1, cc = c + t % 3 - 1;
            m |= (rr >= 0 && rr < 6 && cc >= 0 && cc < 6) ? (1 << t) : 0;
        }
        valid[i] = m;
    }
    const int zero_addr = K::ZERO_OFF + (lane >> 4) * 16;
    for (int i = tid; i < K::STRIDE / 4; i += 256) reinterpret_cast<uint32_t*>(lds + K::ZERO_OFF)[i] = 0u;
    float* pooled = reinterpret_cast<float*>(lds + K::POOL_OFF);
    float* gvec = reinterpret_cast<float*>(lds + K::G_OFF);
    float* hidden = reinterpret_cast<float*>(lds + K::HID_OFF);
    float* plog = reinterpret_cast<float*>(lds + K::PLOG_OFF);
    float* par = reinterpret_cast<float*>(lds + K::PAR_OFF);
    for (int i = tid; i < 5 * kHead; i += 256) {
        float v;
        if (i < kHead) v = fp[P.p_a2 + i];
        else if (i < 2 * kHead) v = fp[P.p_b2 + i - kHead];
        else v = fp[P.p_out + i - 2 * kHead];
        par[i] = v;
    }

    const int64_t n_pass = (N + S - 1) / S;
    for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
        const int64_t n0 = pass * S;
        const int nvalid = (int)((N - n0) < S ? (N - n0) : S);
        __syncthreads();
        // ---- stage the 11 input planes as fp16 rows [cell][32 ch] (ch >= 11 zero) ----
        for (int n = tid; n < K::NPOS; n += 256) {
            const int s = n / 36, p = n - s * 36;
            _Float16 row[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) row[k] = (_Float16)0.f;
            if (s < nvalid) {
                const float* src = planes + (n0 + s) * 396 + p;
#pragma unroll
                for (int ch = 0; ch < 11; ++ch) row[ch] = (_Float16)src[ch * 36];
            }
            h8* dst = reinterpret_cast<h8*>(lds + K::ACT_OFF + n * K::STRIDE);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                h8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = row[q * 8 + k];
                dst[q] = v;
            }
        }
        __syncthreads();

        f4 x[9][4], acc[9][4];
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        // ---- stem: x = relu(conv(planes) + bias) ----
        conv_gemm<C, S, true, true>(x, reinterpret_cast<const h8*>(P.wfrag + P.layer_off[0]), K::CT, ct0, lds, base,
                                    valid, zero_addr, lane);
        {
            const int sub = (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f4 b = *reinterpret_cast<const f4*>(fp + P.stem_bias + chan0 + j * 16 + sub);
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    f4 v = x[i][j] + b;
                    x[i][j] = (f4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                }
            }
        }
        // ---- residual blocks ----
        for (int blk = 0; blk < P.blocks; ++blk) {
            const float* bp = fp + P.blk0 + blk * 3 * C;
            __syncthreads();                                   // everyone finished reading the act buffer
            store_act<C, S, true>(x, lds, tile0, chan0, bp, bp + C, lane);             // t = relu(a1*x + b1)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
            conv_gemm<C, S, true, false>(acc, reinterpret_cast<const h8*>(P.wfrag + P.layer_off[1 + 2 * blk]), K::CT,
                                         ct0, lds, base, valid, zero_addr, lane);
            __syncthreads();
            store_act<C, S, false>(acc, lds, tile0, chan0, nullptr, bp + 2 * C, lane);  // u = relu(conv1 + bias1)
            __syncthreads();
            conv_gemm<C, S, true, false>(x, reinterpret_cast<const h8*>(P.wfrag + P.layer_off[2 + 2 * blk]), K::CT,
                                         ct0, lds, base, valid, zero_addr, lane);       // x += conv2(u)
        }
        // ---- trunk output h = relu(a*x + b) -> LDS; head 1x1 convs (policy 0..63 | value 64..127) ----
        __syncthreads();
        store_act<C, S, true>(x, lds, tile0, chan0, fp + P.trunk_a, fp + P.trunk_b, lane);
        __syncthreads();
        const h8* wh = reinterpret_cast<const h8*>(P.wfrag + P.layer_off[1 + 2 * P.blocks]);
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; x[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; }
        if (K::CG == 1) {
            conv_gemm<C, S, false, false>(acc, wh, 8, 0, lds, base, valid, zero_addr, lane);   // policy map
            conv_gemm<C, S, false, false>(x, wh, 8, 4, lds, base, valid, zero_addr, lane);     // value map
        } else {
            conv_gemm<C, S, false, false>(acc, wh, 8, cg * 4, lds, base, valid, zero_addr, lane);
        }
        __syncthreads();
        // ---- policy head ----
        if (K::CG == 1 || cg == 0) {
            // rows are only 64 channels wide here: reuse the act buffer with the same stride
            const int sub = (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int n = (tile0 + i) * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 b = *reinterpret_cast<const f4*>(fp + P.head_bias + j * 16 + sub);
                    const f4 v = acc[i][j] + b;
                    *reinterpret_cast<h4*>(lds + K::ACT_OFF + n * K::STRIDE + (j * 16 + sub) * 2) =
                        to_h4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
                }
            }
        }
        __syncthreads();
        gpool64<C, S>(lds, pooled, tid);
        __syncthreads();
        for (int it = tid; it < S * kHead; it += 256) {           // g = gpool_linear(pooled)
            const int s = it / kHead, co = it - s * kHead;
            float a = 0.f;
            const float* w = fp + P.p_gwT + co;
            for (int k = 0; k < kPool; ++k) a += w[k * kHead] * pooled[s * kPool + k];
            gvec[it] = a;
        }
        __syncthreads();
        for (int n = tid; n < K::NPOS; n += 256) {                 // three 1x1 output convs on relu(bn2(p + g))
            const int s = n / 36, p = n - s * 36;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
            const unsigned char* row = lds + K::ACT_OFF + n * K::STRIDE;
            for (int c8 = 0; c8 < 8; ++c8) {
                const h8 v = *reinterpret_cast<const h8*>(row + c8 * 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ch = c8 * 8 + k;
                    const float z = fmaxf(((float)v[k] + gvec[s * kHead + ch]) * par[ch] + par[kHead + ch], 0.f);
                    o0 += par[2 * kHead + ch] * z;
                    o1 += par[3 * kHead + ch] * z;
                    o2 += par[4 * kHead + ch] * z;
                }
            }
            plog[(s * 3 + 0) * 36 + p] = o0;
            plog[(s * 3 + 1) * 36 + p] = o1;
            plog[(s * 3 + 2) * 36 + p] = o2;
        }
        __syncthreads();
        if (tid < S * 3) {                                         // log_softmax over the 36 cells
            const int s = tid / 3, h = tid - s * 3;
            if (s < nvalid) {
                const float* v = plog + (s * 3 + h) * 36;
                float mx = -INFINITY;
                for (int p = 0; p < 36; ++p) mx = fmaxf(mx, v[p]);
                float sum = 0.f;
                for (int p = 0; p < 36; ++p) sum += expf(v[p] - mx);
                const float lse = mx + logf(sum);
                float* dst = (h == 0 ? lp1 : h == 1 ? lp2 : lpm) + (n0 + s) * 36;
                for (int p = 0; p < 36; ++p) dst[p] = v[p] - lse;
            }
        }
        __syncthreads();
        // ---- value head ----
        if (K::CG == 1 || cg == 1) {
            const int sub = (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int n = (tile0 + i) * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 b = *reinterpret_cast<const f4*>(fp + P.head_bias + kHead + j * 16 + sub);
                    const f4 v = (K::CG == 1 ? x[i][j] : acc[i][j]) + b;
                    *reinterpret_cast<h4*>(lds + K::ACT_OFF + n * K::STRIDE + (j * 16 + sub) * 2) =
                        to_h4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
                }
            }
        }
        __syncthreads();
        gpool64<C, S>(lds, pooled, tid);
        __syncthreads();
        for (int it = tid; it < S * kMlp; it += 256) {              // fc1 + relu
            const int s = it / kMlp, o = it - s * kMlp;
            float a = fp[P.v_b1 + o];
            const float* w = fp + P.v_w1T + o;
            for (int k = 0; k < kPool; ++k) a += w[k * kMlp] * pooled[s * kPool + k];
            hidden[it] = fmaxf(a, 0.f);
        }
        __syncthreads();
        float* vl = plog;                                           // [S][101] value logits
        for (int it = tid; it < S * kBins; it += 256) {              // fc2
            const int s = it / kBins, o = it - s * kBins;
            float a = fp[P.v_b2 + o];
            const float* w = fp + P.v_w2T + o;
            for (int k = 0; k < kMlp; ++k) a += w[k * kBins] * hidden[s * kMlp + k];
            vl[it] = a;
            if (vlogits != nullptr && s < nvalid) vlogits[(n0 + s) * kBins + o] = a;
        }
        __syncthreads();
        if (tid < S && tid < nvalid && value != nullptr) {           // softmax expectation over bucket centres
            const float* v = vl + tid * kBins;
            float mx = -INFINITY;
            for (int k = 0; k < kBins; ++k) mx = fmaxf(mx, v[k]);
            float sum = 0.f, ex = 0.f;
            for (int k = 0; k < kBins; ++k) {
                const float e = expf(v[k] - mx);
                sum += e;
                ex += e * (-1.0f + 0.02f * (float)k);
            }
            value[n0 + tid] = ex / sum;
        }
    }
}

template <int C, int S>
int launch_net(const NetParams& P, const float* planes, int64_t N, float* lp1, float* lp2, float* lpm,
               float* vlogits, float* value, int max_blocks, hipStream_t st) {
    using K = Cfg<C, S>;
    auto kern = net_forward_kernel<C, S>;
    const int64_t n_pass = (N + S - 1) / S;
    int grid = (int)(n_pass < max_blocks ? n_pass : max_blocks);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), K::LDS_BYTES, st, P, planes, N, lp1, lp2, lpm, vlogits, value);
    return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}

template <int C, int S>
int configure_net() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(net_forward_kernel<C, S>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<C, S>::LDS_BYTES) == hipSuccess
               ? LZ_OK : LZ_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int lz_net_configure(void) {
    int a = configure_net<64, 16>();
    int b = configure_net<128, 8>();
    return a != LZ_OK ? a : b;
}

int lz_net_forward_f16(const LzNetDesc* d, const float* planes, int64_t N, float* lp1, float* lp2, float* lpmc,
                       float* value_logits, float* value, void* stream) {
    if (!d || N < 0) return LZ_ERR_ARG;
    if (N == 0) return LZ_OK;
    if (!d->wfrag || !d->fparams || !planes || !lp1 || !lp2 || !lpmc) return LZ_ERR_ARG;
    if (d->blocks < 0 || d->blocks > 15 || d->num_layers != 2 + 2 * d->blocks) return LZ_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d->wfrag) & 15) || (reinterpret_cast<uintptr_t>(d->fparams) & 15)) return LZ_ERR_ALIGN;
    NetParams P;
    P.wfrag = reinterpret_cast<const _Float16*>(d->wfrag);
    P.fp = d->fparams;
    for (int i = 0; i < d->num_layers; ++i) P.layer_off[i] = d->layer_offsets[i];
    P.blocks = d->blocks;
    P.stem_bias = d->off_stem_bias; P.blk0 = d->off_block0; P.trunk_a = d->off_trunk_a; P.trunk_b = d->off_trunk_b;
    P.head_bias = d->off_head_bias; P.p_gwT = d->off_p_gwT; P.p_a2 = d->off_p_a2; P.p_b2 = d->off_p_b2;
    P.p_out = d->off_p_out; P.v_w1T = d->off_v_w1T; P.v_b1 = d->off_v_b1; P.v_w2T = d->off_v_w2T; P.v_b2 = d->off_v_b2;
    const int max_blocks = d->max_blocks > 0 ? d->max_blocks : 256;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->channels == 64) return launch_net<64, 16>(P, planes, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st);
    if (d->channels == 128) return launch_net<128, 8>(P, planes, N, lp1, lp2, lpmc, value_logits, value, max_blocks, st);
    return LZ_ERR_UNSUPPORTED;
}

}  // extern "C"
