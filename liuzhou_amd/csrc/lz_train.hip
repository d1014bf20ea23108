// lz_train.hip -- fused training loss of the policy + bucketed-value network: forward terms and the gradients with
// respect to the four head outputs in one pass over the batch.
//
// Restates, per sample (one wavefront per sample, lanes over the 220 actions / 101 value bins):
//   combined logits          src/policy_batch.py:95-136  (placement = lp1[cell], movement = lp2[from] + lp1[to],
//                                                         selections = lpmc[cell], auxiliary = 0)
//   masked log-softmax       src/policy_batch.py:139-160 (legal entries only; no legal entry or non-finite lse => 0;
//                                                         non-finite results => -50)
//   policy KL, draw weights  src/policy_batch.py:163-189 (CE - H(target), weight = policy_draw_weight on hard draws,
//                                                         batch value = sum(kl*w) / (sum(w) + 1e-8))
//   two-hot bucket CE        src/neural_network.py:176-198 + v1/python/train_bridge.py:338-352
//   WDL auxiliary term       v1/python/train_bridge.py:30-41,353-358 (weight 0 in the reference: reported, no gradient)
// The PyTorch composition it replaces launches ~45 elementwise / reduction kernels over [B,220] temporaries per step
// (forward + autograd backward); this kernel reads 1.9 KB and writes 0.85 KB per sample once: HBM-bound.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liuzhou_hip.h"
#include "lz_wave.h"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBins = 101;
constexpr int kTotal = 220;

__device__ __forceinline__ int move_dest_cell(int from, int dir) {      // DIRECTIONS: up, down, left, right
    const int r = from / 6, c = from - 6 * r;
    const int nr = r + (dir == 0 ? -1 : dir == 1 ? 1 : 0), nc = c + (dir == 2 ? -1 : dir == 3 ? 1 : 0);
    return (nr < 0 || nr > 5 || nc < 0 || nc > 5) ? -1 : nr * 6 + nc;
}

__global__ __launch_bounds__(kWave * kWavesPerBlock) void policy_value_loss_kernel(
    const float* __restrict__ lp1, const float* __restrict__ lp2, const float* __restrict__ lpm,
    const float* __restrict__ vlogits, const uint8_t* __restrict__ legal, const float* __restrict__ target,
    const float* __restrict__ value, const float* __restrict__ soft, int64_t B, float alpha, float anti_draw,
    float draw_weight, const float* __restrict__ weight_sum, float grad_scale,
    float* __restrict__ terms /*[B,4]: kl, weight, bucket CE, wdl aux*/, float* __restrict__ g1, float* __restrict__ g2,
    float* __restrict__ gm, float* __restrict__ gv) {
    __shared__ float s_dc[kWavesPerBlock][256];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + wv;
    if (s >= B) return;
    float* dc = s_dc[wv];
    const float raw_v = value[s];
    const bool hard_draw = fabsf(raw_v) < 1e-8f;
    const float w = hard_draw ? draw_weight : 1.0f;
    const float wn = grad_scale * w / (weight_sum[0] + 1e-8f);

    // ---- policy ----
    float h1 = 0.f, h2 = 0.f, hm = 0.f;
    if (lane < 36) { h1 = lp1[s * 36 + lane]; h2 = lp2[s * 36 + lane]; hm = lpm[s * 36 + lane]; }
    float c[4], t[4]; bool lg[4];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int a = it * kWave + lane;
        const bool in = a < kTotal;
        int from = 0, dest = 0, cell = 0;
        bool dest_ok = true;
        if (a >= 36 && a < 180) {
            from = (a - 36) >> 2;
            const int d = move_dest_cell(from, (a - 36) & 3);
            dest_ok = d >= 0;
            dest = d < 0 ? 0 : d;
        } else if (a >= 180 && a < 216) cell = a - 180;
        else if (a < 36) cell = a;
        const float p1d = __shfl(h1, dest), p2f = __shfl(h2, from), p1c = __shfl(h1, cell), pmc = __shfl(hm, cell);
        float x = a < 36 ? p1c : a < 180 ? (dest_ok ? p2f + p1d : -INFINITY) : a < 216 ? pmc : 0.f;
        lg[it] = in && legal[s * kTotal + (in ? a : 0)] != 0;
        t[it] = in ? target[s * kTotal + a] : 0.f;
        c[it] = lg[it] ? x : -INFINITY;
        mx = fmaxf(mx, c[it]);
    }
    mx = lzw::wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) se += (lg[it] && c[it] > -INFINITY) ? expf(c[it] - mx) : 0.f;
    se = lzw::wave_sum(se);
    const uint64_t any_legal = __ballot(lg[0]) | __ballot(lg[1]) | __ballot(lg[2]) | __ballot(lg[3]);
    float lse = mx + logf(se);
    const bool lse_ok = any_legal != 0 && isfinite(lse);
    if (!lse_ok) lse = 0.f;
    float ce = 0.f, ent = 0.f, gsum = 0.f;
    float gl[4], p[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        float lp = lg[it] ? c[it] - lse : 0.f;
        const bool fin = isfinite(lp);
        if (!fin) lp = -50.f;
        const bool pass = lg[it] && fin && lp >= -50.f;                 // gradient gate of where() and clamp(min=-50)
        const float lps = fmaxf(lp, -50.f);
        ce -= t[it] * lps;
        ent -= t[it] * logf(fmaxf(t[it], 1e-8f));
        gl[it] = pass ? -t[it] * wn : 0.f;                               // dL / d log_prob
        gsum += gl[it];
        p[it] = (lse_ok && lg[it] && c[it] > -INFINITY) ? expf(c[it] - lse) : 0.f;
    }
    ce = lzw::wave_sum(ce);
    ent = lzw::wave_sum(ent);
    gsum = lzw::wave_sum(gsum);
#pragma unroll
    for (int it = 0; it < 4; ++it) dc[it * kWave + lane] = gl[it] - (lse_ok ? p[it] * gsum : 0.f);   // dL / d combined
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 36) {
        // head gradients: lp1 collects its placement entry and every move that lands on the cell
        float a1 = dc[lane];
        const int r = lane / 6, col = lane - 6 * r;
        if (r < 5) a1 += dc[36 + (lane + 6) * 4 + 0];       // up from the cell below
        if (r > 0) a1 += dc[36 + (lane - 6) * 4 + 1];       // down from the cell above
        if (col < 5) a1 += dc[36 + (lane + 1) * 4 + 2];     // left from the right neighbour
        if (col > 0) a1 += dc[36 + (lane - 1) * 4 + 3];     // right from the left neighbour
        const float a2 = dc[36 + lane * 4] + dc[37 + lane * 4] + dc[38 + lane * 4] + dc[39 + lane * 4];
        g1[s * 36 + lane] = a1;
        g2[s * 36 + lane] = a2;
        gm[s * 36 + lane] = dc[180 + lane];
    }

    // ---- value: two-hot bucket cross entropy on clamp((1-alpha)*v + alpha*soft, -1, 1) ----
    float v_used = raw_v;
    if (fabsf(anti_draw) > 1e-9f && hard_draw) v_used = anti_draw;
    float mixed = (1.0f - alpha) * v_used + alpha * soft[s];
    mixed = fminf(fmaxf(mixed, -1.0f), 1.0f);
    const float step = 2.0f / (float)(kBins - 1);
    const float u = (mixed + 1.0f) / step;
    int lo = (int)floorf(u);
    lo = lo < 0 ? 0 : (lo > kBins - 1 ? kBins - 1 : lo);
    const int hi = lo + 1 > kBins - 1 ? kBins - 1 : lo + 1;
    float frac = fminf(fmaxf(u - (float)lo, 0.f), 1.f);
    if (hi == lo) frac = 0.f;
    const float v0 = vlogits[s * kBins + lane];
    const bool has1 = lane + 64 < kBins;
    const float v1 = has1 ? vlogits[s * kBins + lane + 64] : -INFINITY;
    const float vm = lzw::wave_max(fmaxf(v0, v1));
    const float e0 = expf(v0 - vm), e1 = has1 ? expf(v1 - vm) : 0.f;
    const float vs = lzw::wave_sum(e0 + e1);
    const float vlse = vm + logf(vs);
    const float tg0 = (lane == lo ? 1.0f - frac : 0.f) + (lane == hi ? frac : 0.f);
    const float tg1 = (lane + 64 == lo ? 1.0f - frac : 0.f) + (lane + 64 == hi ? frac : 0.f);
    float cev = -(tg0 * (v0 - vlse)) - (has1 ? tg1 * (v1 - vlse) : 0.f);
    cev = lzw::wave_sum(cev);
    const float sm0 = e0 / vs, sm1 = e1 / vs;
    const float gvs = grad_scale / (float)B;
    gv[s * kBins + lane] = (sm0 - tg0) * gvs;
    if (has1) gv[s * kBins + lane + 64] = (sm1 - tg1) * gvs;
    // WDL auxiliary term (reported only).  Bins are classed by torch.linspace(-1, 1, 101) against +-1e-8
    // (train_bridge.py:33-36); its float32 centre bin is 2.2e-8, i.e. a WIN bin, so the draw class is empty.
    const float pw = lzw::wave_sum((lane >= 50 ? sm0 : 0.f) + (has1 ? sm1 : 0.f));
    const float pd = 0.f;
    const float pl = lzw::wave_sum(lane < 50 ? sm0 : 0.f);
    const float wsum3 = fmaxf(pw + pd + pl, 1e-8f);
    const float tw = fmaxf(raw_v, 0.f), tl = fmaxf(-raw_v, 0.f), td = fmaxf(1.0f - tw - tl, 0.f);
    const float aux = -(tw * logf(fmaxf(pw / wsum3, 1e-8f)) + td * logf(fmaxf(pd / wsum3, 1e-8f)) +
                        tl * logf(fmaxf(pl / wsum3, 1e-8f)));
    if (lane == 0) {
        terms[s * 4 + 0] = ce - ent;
        terms[s * 4 + 1] = w;
        terms[s * 4 + 2] = cev;
        terms[s * 4 + 3] = aux;
    }
}

}  // namespace

extern "C" int lz_policy_value_loss_fwd_bwd(const float* log_p1, const float* log_p2, const float* log_pmc,
                                            const float* value_logits, const uint8_t* legal_mask,
                                            const float* policy_target, const float* value_target,
                                            const float* soft_value_target, int64_t batch, float soft_label_alpha,
                                            float anti_draw_penalty, float policy_draw_weight,
                                            const float* policy_weight_sum, float grad_scale, float* terms,
                                            float* grad_log_p1, float* grad_log_p2, float* grad_log_pmc,
                                            float* grad_value_logits, void* stream) {
    if (batch < 0) return LZ_ERR_ARG;
    if (batch == 0) return LZ_OK;
    if (!log_p1 || !log_p2 || !log_pmc || !value_logits || !legal_mask || !policy_target || !value_target ||
        !soft_value_target || !policy_weight_sum || !terms || !grad_log_p1 || !grad_log_p2 || !grad_log_pmc ||
        !grad_value_logits)
        return LZ_ERR_ARG;
    const unsigned grid = (unsigned)((batch + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(policy_value_loss_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), log_p1, log_p2, log_pmc, value_logits, legal_mask,
                       policy_target, value_target, soft_value_target, batch, soft_label_alpha, anti_draw_penalty,
                       policy_draw_weight, policy_weight_sum, grad_scale, terms, grad_log_p1, grad_log_p2, grad_log_pmc,
                       grad_value_logits);
    return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}

// ---- compact trajectory records (wire format of the per-iteration gather) ----------------
// A trajectory row of the reference contract is 2 692 B (f32[11,6,6] planes + bool[220] + f32[220] + 2 f32); its
// information content is far smaller: the planes are 0/1 (4 bitboards + a phase), the policy is non-zero only on the
// <= 72 legal entries.  Record = 360 B, exact (pack -> unpack reproduces every byte of the five tensors):
//   u64 w[4]   own | phase << 36, opp, own-marked, opp-marked        (planes 0..3 as bits, planes 4..10 one-hot)
//   u32 m[7]   legal mask bits 0..219
//   f32 p[72]  policy of the legal actions in ascending index order
//   f32 v[2]   value target, soft value target            (+ 4 B pad)
namespace {
constexpr int kRecBytes = 360;

__global__ __launch_bounds__(kWave * kWavesPerBlock) void pack_rows_kernel(
    const float* __restrict__ planes, const uint8_t* __restrict__ legal, const float* __restrict__ policy,
    const float* __restrict__ value, const float* __restrict__ soft, int64_t n, uint8_t* __restrict__ out,
    int* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (s >= n) return;
    uint8_t* rec = out + s * kRecBytes;
    const float* pl = planes + s * 396;
    uint64_t w[4];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x = lane < 36 ? pl[p * 36 + lane] : 0.f;
        ok = ok && (x == 0.f || x == 1.f);
        w[p] = __ballot(x != 0.f);
    }
    int phase = 0;
#pragma unroll
    for (int ph = 1; ph <= 7; ++ph) {
        const float x = lane < 36 ? pl[(3 + ph) * 36 + lane] : 0.f;
        const uint64_t b = __ballot(lane < 36 && x != 0.f);
        ok = ok && (x == 0.f || x == 1.f) && (b == 0 || b == 0xFFFFFFFFFull);
        if (b) { ok = ok && phase == 0; phase = ph; }
    }
    uint32_t mw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int base = 0;
    float* pv = reinterpret_cast<float*>(rec + 60);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int a = it * kWave + lane;
        const bool in = a < kTotal;
        const bool lg = in && legal[s * kTotal + (in ? a : 0)] != 0;
        const float p = in ? policy[s * kTotal + a] : 0.f;
        ok = ok && (lg || p == 0.f);
        const uint64_t bal = __ballot(lg);
        mw[2 * it] = (uint32_t)bal; mw[2 * it + 1] = (uint32_t)(bal >> 32);
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (lg && slot < 72) pv[slot] = p;
        base += __popcll(bal);
    }
    ok = ok && base <= 72;
    for (int k = base + lane; k < 72; k += kWave) pv[k] = 0.f;
    if (lane == 0) {
        uint64_t* wq = reinterpret_cast<uint64_t*>(rec);
        wq[0] = w[0] | ((uint64_t)phase << 36); wq[1] = w[1]; wq[2] = w[2]; wq[3] = w[3];
        uint32_t* mq = reinterpret_cast<uint32_t*>(rec + 32);
#pragma unroll
        for (int k = 0; k < 7; ++k) mq[k] = mw[k];
        float* vq = reinterpret_cast<float*>(rec + 348);
        vq[0] = value[s]; vq[1] = soft[s]; vq[2] = 0.f;
    }
    if (__ballot(!ok) != 0 && lane == 0) atomicAdd(bad, 1);     // row not representable (never for self-play output)
}

__global__ __launch_bounds__(kWave * kWavesPerBlock) void unpack_rows_kernel(
    const uint8_t* __restrict__ in, int64_t n, float* __restrict__ planes, uint8_t* __restrict__ legal,
    float* __restrict__ policy, float* __restrict__ value, float* __restrict__ soft) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (s >= n) return;
    const uint8_t* rec = in + s * kRecBytes;
    const uint64_t* wq = reinterpret_cast<const uint64_t*>(rec);
    const uint32_t* mq = reinterpret_cast<const uint32_t*>(rec + 32);
    const float* pv = reinterpret_cast<const float*>(rec + 60);
    const uint64_t w0 = wq[0];
    const int phase = (int)((w0 >> 36) & 7);
    float* pl = planes + s * 396;
    for (int j = lane; j < 396; j += kWave) {
        const int p = j / 36, c = j - p * 36;
        const bool bit = p < 4 ? ((p == 0 ? w0 : wq[p]) >> c) & 1 : (phase == p - 3);
        pl[j] = bit ? 1.f : 0.f;
    }
    int base = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int a = it * kWave + lane;
        const uint64_t bal = (uint64_t)mq[2 * it] | ((2 * it + 1 < 7) ? ((uint64_t)mq[2 * it + 1] << 32) : 0ull);
        const bool lg = (bal >> lane) & 1;
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (a < kTotal) {
            legal[s * kTotal + a] = lg ? 1 : 0;
            policy[s * kTotal + a] = lg ? pv[slot] : 0.f;
        }
        base += __popcll(bal);
    }
    if (lane == 0) {
        const float* vq = reinterpret_cast<const float*>(rec + 348);
        value[s] = vq[0]; soft[s] = vq[1];
    }
}
}  // namespace

extern "C" int lz_pack_trajectory_rows(const float* state_tensors, const uint8_t* legal_masks,
                                       const float* policy_targets, const float* value_targets,
                                       const float* soft_value_targets, int64_t rows, void* records,
                                       int32_t* not_representable, void* stream) {
    if (rows < 0) return LZ_ERR_ARG;
    if (rows == 0) return LZ_OK;
    if (!state_tensors || !legal_masks || !policy_targets || !value_targets || !soft_value_targets || !records ||
        !not_representable)
        return LZ_ERR_ARG;
    const unsigned grid = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), state_tensors, legal_masks, policy_targets, value_targets,
                       soft_value_targets, rows, reinterpret_cast<uint8_t*>(records), not_representable);
    return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}

extern "C" int lz_unpack_trajectory_rows(const void* records, int64_t rows, float* state_tensors, uint8_t* legal_masks,
                                         float* policy_targets, float* value_targets, float* soft_value_targets,
                                         void* stream) {
    if (rows < 0) return LZ_ERR_ARG;
    if (rows == 0) return LZ_OK;
    if (!records || !state_tensors || !legal_masks || !policy_targets || !value_targets || !soft_value_targets)
        return LZ_ERR_ARG;
    const unsigned grid = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint8_t*>(records), rows,
                       state_tensors, legal_masks, policy_targets, value_targets, soft_value_targets);
    return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH;
}
