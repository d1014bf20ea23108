// lz_ops.hip -- gfx950 kernels behind the reference's v0_core operator surface (C ABI in
// include/liuzhou_hip.h).  Hand-written for CDNA4: 64-lane wavefronts, one wave per game state for
// the wide-output operators (lane = action slot / output vector, bitboards built with one ballot),
// one lane per state for the transition operators (a whole state lives in registers as bitboards).
//
// All of these are HBM-bound byte/integer kernels; the design goal is full-width coalesced stores
// (16 B per lane) and no intermediate tensors.  No MFMA here by construction.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "lz_soa.h"
#include "lz_wave.h"

using namespace lz;

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH; }
inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// wave-uniform state index (kept in an SGPR so the per-state scalars come through scalar loads)
__device__ __forceinline__ int64_t wave_item() {
    int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    return (int64_t)blockIdx.x * kWavesPerBlock + w;
}

// ---- wave-level helpers -------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(v, o);
        v = other > v ? other : v;
    }
    return v;
}
// order-preserving float -> uint32 (NaN must be filtered by the caller)
__device__ __forceinline__ uint32_t float_order(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// per-wave load of one SoA row as bitboards: lanes 0..35 read one cell each, one ballot per set
struct RowBits { uint64_t black, white, empty, mb, mw; };
__device__ __forceinline__ RowBits load_row_bits(const int8_t* board, const uint8_t* mb, const uint8_t* mw,
                                                 int64_t st, int lane) {
    int bv = 2; int b1 = 0, b2 = 0;
    if (lane < kCells) {
        bv = board[st * kCells + lane];
        b1 = mb[st * kCells + lane];
        b2 = mw[st * kCells + lane];
    }
    RowBits r;
    r.black = __ballot(bv == 1);
    r.white = __ballot(bv == -1);
    r.empty = __ballot(bv == 0);
    r.mb = __ballot(b1 != 0);
    r.mw = __ballot(b2 != 0);
    return r;
}
__device__ __forceinline__ uint64_t pick(const RowBits& r, int v) {
    return v == 1 ? r.black : v == -1 ? r.white : v == 0 ? r.empty : 0ull;
}

// =================================================================================================
// encode_actions_fast: one wave per state, lane = action slot.  Reads 180 B, writes T + 16 T bytes.
// =================================================================================================
template <bool PACKED_MASK>
__global__ __launch_bounds__(kBlock) void encode_actions_kernel(LzStateSoA s, int64_t B, int T,
                                                                uint8_t* __restrict__ mask,
                                                                int32_t* __restrict__ meta) {
    const int lane = lane_id();
    const int64_t st = wave_item();
    if (st >= B) return;
    const RowBits rb = load_row_bits(s.board, s.marks_black, s.marks_white, st, lane);
    const int phase = (int)s.phase[st];
    const int cur = (int)s.current_player[st];
    const int pm_rem = (int)s.pending_marks_remaining[st];
    const int pc_rem = (int)s.pending_captures_remaining[st];
    const int forced = (int)s.forced_removals_done[st];
    const Legal L = legal_actions(rb.black, rb.white, pick(rb, cur), pick(rb, -cur), rb.empty, rb.mb, rb.mw,
                                  phase, cur, pm_rem, pc_rem, forced, /*fallback_forced=*/1);
    int4* mrow = reinterpret_cast<int4*>(meta) + st * T;
    uint8_t* krow = mask + st * T;
    const int iters = (T + kWave - 1) / kWave;
    for (int it = 0; it < iters; ++it) {
        const int a = it * kWave + lane;
        int kind = -1, p = -1, q = -1, e = -1;
        bool lg = false;
        if (a < 36) {
            lg = (L.place >> a) & 1;
            if (lg) { kind = kActPlace; p = a; }
        } else if (a < 180) {
            const int from = (a - 36) >> 2, d = (a - 36) & 3;
            lg = (move_set(L, d) >> from) & 1;
            if (lg) { kind = kActMove; p = from; q = d; e = move_dest(from, d); }
        } else if (a < 216) {
            const int c = a - 180;
            lg = (L.sel >> c) & 1;
            if (lg) { kind = L.sel_kind; p = c; }
        } else if (a == 216) {
            lg = L.process != 0;
            if (lg) kind = kActProcess;
        }
        const bool in_range = a < T;
        lg = lg && in_range;
        if (in_range) mrow[a] = make_int4(kind, p, q, e);
        if (PACKED_MASK) {
            const uint64_t bal = __ballot(lg);
            const int a4 = it * kWave + lane * 4;
            if (lane < 16 && a4 < T) {
                const uint32_t nib = (uint32_t)(bal >> (lane * 4)) & 0xFu;
                *reinterpret_cast<uint32_t*>(krow + a4) = (nib * 0x00204081u) & 0x01010101u;
            }
        } else {
            if (in_range) krow[a] = lg ? 1 : 0;
        }
    }
}

// =================================================================================================
// batch_apply_moves: one lane per (action, parent).  The parent row is gathered as 27 dwords (children
// of one parent are adjacent, so these hit L1/L2), the move is applied on bitboards in registers and
// the child row is written back as dwords.
// =================================================================================================
__device__ __forceinline__ void load_row_words(const void* base, int64_t row, uint32_t (&w)[9]) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(base) + row * kCells);
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = p[i];
}
__device__ __forceinline__ void store_row_words(void* base, int64_t row, const uint32_t (&w)[9]) {
    uint32_t* p = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(base) + row * kCells);
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = w[i];
}
__device__ __forceinline__ uint64_t words_eq(const uint32_t (&w)[9], uint32_t byte_value) {
    uint64_t m = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= (uint64_t)(((w[i] >> (8 * k)) & 0xFFu) == byte_value) << (i * 4 + k);
    }
    return m;
}
__device__ __forceinline__ uint64_t words_nonzero(const uint32_t (&w)[9]) {
    uint64_t m = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= (uint64_t)(((w[i] >> (8 * k)) & 0xFFu) != 0u) << (i * 4 + k);
    }
    return m;
}
__device__ __forceinline__ void board_words(uint64_t black, uint64_t white, uint32_t (&w)[9]) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = i * 4 + k;
            const uint32_t byte = ((black >> c) & 1) ? 0x01u : (((white >> c) & 1) ? 0xFFu : 0u);
            v |= byte << (8 * k);
        }
        w[i] = v;
    }
}
__device__ __forceinline__ void mark_words(uint64_t m, uint32_t (&w)[9]) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t nib = (uint32_t)(m >> (i * 4)) & 0xFu;
        w[i] = (nib * 0x00204081u) & 0x01010101u;
    }
}

struct LoadedState { State s; uint64_t other; int64_t raw_player; };   // `other` = cells that are neither -1/0/1

__device__ __forceinline__ State load_state(const LzStateSoA& s, int64_t row) {
    uint32_t w[9];
    State st;
    load_row_words(s.board, row, w);
    st.black = words_eq(w, 0x01u);
    st.white = words_eq(w, 0xFFu);
    load_row_words(s.marks_black, row, w);
    st.mb = words_nonzero(w);
    load_row_words(s.marks_white, row, w);
    st.mw = words_nonzero(w);
    st.phase = (int)s.phase[row];
    st.player = (int)s.current_player[row];
    st.pm_req = (int)s.pending_marks_required[row];
    st.pm_rem = (int)s.pending_marks_remaining[row];
    st.pc_req = (int)s.pending_captures_required[row];
    st.pc_rem = (int)s.pending_captures_remaining[row];
    st.forced = (int)s.forced_removals_done[row];
    st.move_count = (int)s.move_count[row];
    st.msc = (int)s.moves_since_capture[row];
    return st;
}
__device__ __forceinline__ void store_state(const LzStateSoA& o, int64_t row, const State& st) {
    uint32_t w[9];
    board_words(st.black, st.white, w);
    store_row_words(o.board, row, w);
    mark_words(st.mb, w);
    store_row_words(o.marks_black, row, w);
    mark_words(st.mw, w);
    store_row_words(o.marks_white, row, w);
    o.phase[row] = st.phase;
    o.current_player[row] = st.player;
    o.pending_marks_required[row] = st.pm_req;
    o.pending_marks_remaining[row] = st.pm_rem;
    o.pending_captures_required[row] = st.pc_req;
    o.pending_captures_remaining[row] = st.pc_rem;
    o.forced_removals_done[row] = st.forced;
    o.move_count[row] = st.move_count;
    o.moves_since_capture[row] = st.msc;
}

__global__ __launch_bounds__(kBlock) void apply_moves_kernel(LzStateSoA in, int64_t B,
                                                             const int4* __restrict__ codes,
                                                             const int64_t* __restrict__ parents,
                                                             int64_t N, LzStateSoA out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const int64_t p = parents[i];
    if (p < 0 || p >= B) return;
    const int4 code = codes[i];
    State st = load_state(in, p);
    apply(st, code.x, code.y, code.z);
    store_state(out, i, st);
}

__global__ __launch_bounds__(kBlock) void apply_moves_inplace_kernel(LzStateSoA s, int64_t B,
                                                                     const int4* __restrict__ codes,
                                                                     const int64_t* __restrict__ slots,
                                                                     int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const int64_t p = slots[i];
    if (p < 0 || p >= B) return;
    const int4 code = codes[i];
    State st = load_state(s, p);
    apply(st, code.x, code.y, code.z);
    store_state(s, p, st);
}

// =================================================================================================
// states_to_model_input: one wave per state; 11*36 floats = 99 float4, lane = output float4.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void model_input_kernel(const int8_t* __restrict__ board,
                                                             const uint8_t* __restrict__ mb,
                                                             const uint8_t* __restrict__ mw,
                                                             const int64_t* __restrict__ phase,
                                                             const int64_t* __restrict__ player, int64_t B,
                                                             float* __restrict__ out) {
    const int lane = lane_id();
    const int64_t st = wave_item();
    if (st >= B) return;
    const int64_t cur64 = player[st];
    const int cur8 = (int)(int8_t)cur64;                 // encoding.cpp:48 casts the player to int8
    const int neg8 = (int)(int8_t)(-(int8_t)cur64);
    int bv = 0x7FFF; int b1 = 0, b2 = 0;
    if (lane < kCells) {
        bv = board[st * kCells + lane];
        b1 = mb[st * kCells + lane];
        b2 = mw[st * kCells + lane];
    }
    const uint64_t own = __ballot(bv == cur8);
    const uint64_t opp = __ballot(bv == neg8);
    const uint64_t m1 = __ballot(b1 != 0), m2 = __ballot(b2 != 0);
    const bool is_black = cur64 == 1;
    const uint64_t self_marks = is_black ? m1 : m2;
    const uint64_t opp_marks = is_black ? m2 : m1;
    const int64_t ph = phase[st];
    float4* orow = reinterpret_cast<float4*>(out + st * 11 * kCells);
    for (int j = lane; j < 99; j += kWave) {
        const int plane = j / 9;
        const int cell0 = (j - plane * 9) * 4;
        uint32_t bits;
        if (plane < 4) {
            const uint64_t src = plane == 0 ? own : plane == 1 ? opp : plane == 2 ? self_marks : opp_marks;
            bits = (uint32_t)(src >> cell0) & 0xFu;
        } else {
            bits = (ph == (int64_t)(plane - 3)) ? 0xFu : 0u;
        }
        orow[j] = make_float4((bits & 1) ? 1.f : 0.f, (bits & 2) ? 1.f : 0.f, (bits & 4) ? 1.f : 0.f,
                              (bits & 8) ? 1.f : 0.f);
    }
}

// =================================================================================================
// project_policy_logits_fast: one wave per row; the three 36-wide heads live in lanes 0..35 and are
// gathered with wave shuffles; masked softmax over T <= 256 columns with wave reductions.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void project_policy_kernel(const float* __restrict__ lp1,
                                                                const float* __restrict__ lp2,
                                                                const float* __restrict__ lpmc,
                                                                const uint8_t* __restrict__ mask, int64_t B,
                                                                int T, float* __restrict__ probs,
                                                                float* __restrict__ masked_logits) {
    const int lane = lane_id();
    const int64_t row = wave_item();
    if (row >= B) return;
    float h1 = 0.f, h2 = 0.f, hm = 0.f;
    if (lane < kCells) {
        h1 = lp1[row * kCells + lane];
        h2 = lp2[row * kCells + lane];
        hm = lpmc[row * kCells + lane];
    }
    const float ninf = -INFINITY;
    float v[4];
    bool lg[4];
    float mx = ninf;
    bool any_legal = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int a = it * kWave + lane;
        // shuffles are executed by every lane (uniform control flow)
        int from = 0, dest = 0, cell = 0;
        bool dest_ok = false;
        if (a >= 36 && a < 180) {
            from = (a - 36) >> 2;
            const int d = (a - 36) & 3;
            const int r = from / 6, c = from - 6 * r;
            dest_ok = !((d == 0 && r == 0) || (d == 1 && r == 5) || (d == 2 && c == 0) || (d == 3 && c == 5));
            dest = dest_ok ? move_dest(from, d) : 0;
        } else if (a >= 180 && a < 216) {
            cell = a - 180;
        } else if (a < 36) {
            cell = a;
        }
        const float p1_dest = __shfl(h1, dest);
        const float p2_from = __shfl(h2, from);
        const float p1_cell = __shfl(h1, cell);
        const float pm_cell = __shfl(hm, cell);
        float x;
        if (a < 36) x = p1_cell;
        else if (a < 180) x = dest_ok ? (p2_from + p1_dest) : ninf;
        else if (a < 216) x = pm_cell;
        else x = 0.f;
        const bool legal = (a < T) && mask[row * T + (a < T ? a : 0)] != 0;
        v[it] = legal ? x : ninf;
        lg[it] = legal;
        any_legal = any_legal || legal;
        if (v[it] > mx) mx = v[it];          // NaN never becomes the max (matches isfinite gating below)
    }
    any_legal = __ballot(any_legal) != 0ull;
    bool fin = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) fin = fin || isfinite(v[it]);
    const bool any_finite = __ballot(fin) != 0ull;
    mx = wave_max(mx);
    float e[4];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        e[it] = (v[it] == ninf) ? 0.f : expf(v[it] - mx);
        sum += e[it];
    }
    sum = wave_sum(sum);
    const bool do_softmax = any_legal && any_finite;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int a = it * kWave + lane;
        if (a >= T) continue;
        float pr = 0.f, ml = v[it];
        if (do_softmax) pr = e[it] / sum;
        else if (any_legal && lg[it]) ml = 0.f;      // project_policy_logits_fast.cpp:153-160
        probs[row * T + a] = pr;
        masked_logits[row * T + a] = ml;
    }
}

// =================================================================================================
// root_pack_rows: one wave per row; ballot + popcount prefix gives each legal action its packed slot.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void root_pack_kernel(const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ probs,
                                                           const int4* __restrict__ meta, int64_t B, int T,
                                                           int cap, int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ legal_index,
                                                           float* __restrict__ priors,
                                                           int4* __restrict__ codes) {
    const int lane = lane_id();
    const int64_t row = wave_item();
    if (row >= B) return;
    const int iters = (T + kWave - 1) / kWave;
    // pass 1: row sum of legal probabilities (module.cpp:337: priors / sum.clamp_min(1e-8))
    float part = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int a = it * kWave + lane;
        if (a < T && mask[row * T + a]) part += probs[row * T + a];
    }
    const float denom = fmaxf(wave_sum(part), 1e-8f);
    // clear the row's padding
    for (int j = lane; j < cap; j += kWave) {
        legal_index[row * cap + j] = -1;
        priors[row * cap + j] = 0.f;
        codes[row * cap + j] = make_int4(0, 0, 0, 0);
    }
    int base = 0;
    for (int it = 0; it < iters; ++it) {
        const int a = it * kWave + lane;
        const bool lg = a < T && mask[row * T + a] != 0;
        const uint64_t bal = __ballot(lg);
        if (lg) {
            const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (slot < cap) {
                legal_index[row * cap + slot] = a;
                priors[row * cap + slot] = probs[row * T + a] / denom;
                codes[row * cap + slot] = meta[row * T + a];
            }
        }
        base += __popcll(bal);
    }
    if (lane == 0) counts[row] = base;
}

// root_pack_sparse_actions, second half (module.cpp:285-363): the reference's data-dependent [R, Amax] outputs from the
// fixed-capacity rows.  Plan: one workgroup scans the counts -- rank of every non-terminal row among the valid roots,
// exclusive sum of the counts (offset of the row's children in the flat lists), R / Amax / N.  Fill: one wave per row.
constexpr int kPlanBlock = 1024;
__global__ __launch_bounds__(kPlanBlock) void root_pack_plan_kernel(const int32_t* __restrict__ counts, int64_t B,
                                                                    int32_t* __restrict__ rank,
                                                                    int64_t* __restrict__ child_off,
                                                                    int64_t* __restrict__ sizes) {
    __shared__ int s_roots[kPlanBlock / kWave];
    __shared__ long long s_kids[kPlanBlock / kWave];
    __shared__ int s_max[kPlanBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int64_t per = (B + kPlanBlock - 1) / kPlanBlock;
    const int64_t lo = tid * per, hi = (lo + per < B) ? lo + per : B;
    int roots = 0, mx = 0;
    long long kids = 0;
    for (int64_t j = lo; j < hi; ++j) {
        const int c = counts[j];
        roots += c > 0 ? 1 : 0;
        kids += c > 0 ? c : 0;
        mx = c > mx ? c : mx;
    }
    int ri = roots;
    long long ki = kids;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int rv = __shfl_up(ri, d, kWave);
        const long long kv = __shfl_up(ki, d, kWave);
        if (lane >= d) { ri += rv; ki += kv; }
    }
    int wmx = mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(wmx, o); wmx = v > wmx ? v : wmx; }
    if (lane == kWave - 1) { s_roots[w] = ri; s_kids[w] = ki; }
    if (lane == 0) s_max[w] = wmx;
    __syncthreads();
    int rb = 0, rt = 0, amax = 0;
    long long kb = 0, kt = 0;
    for (int i = 0; i < kPlanBlock / kWave; ++i) {
        if (i < w) { rb += s_roots[i]; kb += s_kids[i]; }
        rt += s_roots[i]; kt += s_kids[i];
        amax = s_max[i] > amax ? s_max[i] : amax;
    }
    int r = rb + ri - roots;
    long long k = kb + ki - kids;
    for (int64_t j = lo; j < hi; ++j) {
        const int c = counts[j];
        rank[j] = c > 0 ? r : -1;
        child_off[j] = k;
        if (c > 0) { ++r; k += c; }
    }
    if (tid == 0) { sizes[0] = rt; sizes[1] = amax; sizes[2] = kt; }
}

__global__ __launch_bounds__(kBlock) void root_pack_fill_kernel(
    const int32_t* __restrict__ counts, const int32_t* __restrict__ legal_index, const float* __restrict__ priors,
    const int4* __restrict__ codes, const int32_t* __restrict__ rank, const int64_t* __restrict__ child_off, int64_t B,
    int cap, int M, uint8_t* __restrict__ terminal_mask, int64_t* __restrict__ valid_root_indices,
    int64_t* __restrict__ counts_out, uint8_t* __restrict__ valid_mask, int64_t* __restrict__ legal_index_mat,
    float* __restrict__ priors_mat, int4* __restrict__ action_code_mat, int64_t* __restrict__ pack_flat_idx,
    int4* __restrict__ action_codes_all, int64_t* __restrict__ parent_indices_all) {
    const int lane = lane_id();
    const int64_t b = wave_item();
    if (b >= B) return;
    const int c = counts[b];
    const int r = rank[b];
    if (lane == 0) terminal_mask[b] = c == 0 ? 1 : 0;
    if (r < 0) return;
    if (lane == 0) { valid_root_indices[r] = b; counts_out[r] = c; }
    const int64_t off = child_off[b];
    for (int k = lane; k < M; k += kWave) {
        const bool ok = k < c && k < cap;
        const int64_t o = (int64_t)r * M + k;
        valid_mask[o] = ok ? 1 : 0;
        const int li = ok ? legal_index[b * cap + k] : 0;
        legal_index_mat[o] = li < 0 ? 0 : li;                       // module.cpp:327: clamp_min(0) on the padding
        priors_mat[o] = ok ? priors[b * cap + k] : 0.f;
        const int4 code = ok ? codes[b * cap + k] : make_int4(0, 0, 0, 0);
        action_code_mat[o] = code;
        if (ok) {
            pack_flat_idx[off + k] = o;
            action_codes_all[off + k] = code;
            parent_indices_all[off + k] = b;
        }
    }
}

// =================================================================================================
// root_puct_allocate_visits: one wave per root, statistics in registers for all simulations.
// SLOTS actions per lane (A <= 64*SLOTS).  Each pull is a single 64-bit wave max over
// (order(score) << 32 | ~index): highest score, lowest index on ties -- no LDS, no barrier.
// =================================================================================================
template <int SLOTS>
__global__ __launch_bounds__(kBlock) void root_puct_kernel(const float* __restrict__ priors,
                                                           const float* __restrict__ leaf,
                                                           const uint8_t* __restrict__ valid, int64_t R,
                                                           int A, int64_t sims, float c,
                                                           float* __restrict__ visits,
                                                           float* __restrict__ value_sum,
                                                           float* __restrict__ root_values) {
    const int lane = lane_id();
    const int64_t root = wave_item();
    if (root >= R) return;
    float cp[SLOTS], lv[SLOTS], vis[SLOTS], vs[SLOTS], q[SLOTS];
    bool ok[SLOTS];
    int live = 0;                                       // wave-uniform: slots beyond the last valid action are skipped
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int a = j * kWave + lane;
        ok[j] = a < A && valid[root * A + (a < A ? a : 0)] != 0;
        cp[j] = c * (a < A ? priors[root * A + a] : 0.f);           // first product of c * p * sqrt_total
        lv[j] = a < A ? leaf[root * A + a] : 0.f;
        vis[j] = 0.f;
        vs[j] = 0.f;
        q[j] = 0.f;                                     // vs / max(vis, 1e-8), refreshed when the action is visited
        if (__ballot(ok[j]) != 0ull) live = j + 1;
    }
    float total = 0.f;
    for (int64_t sim = 0; sim < sims; ++sim) {
        const float sqrt_total = sqrtf(total + 1.0f);      // correctly rounded (hipcc default)
        // same operation order as root_puct_fused.cu:53-56; built with -ffp-contract=off.  Highest score, lowest
        // index on ties: wave maximum of the scores on DPP, then the first lane (of the first slot) that holds it.
        float sc[SLOTS];
        float best = -INFINITY;
        bool any = false;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            sc[j] = -INFINITY;
            if (j < live) {
                const float u = cp[j] * sqrt_total / (1.0f + vis[j]);
                const float score = (q[j] + u) + 0.0f;                // +0 canonicalises -0
                const bool cand = ok[j] && score == score;
                if (cand) { sc[j] = score; any = true; best = score > best ? score : best; }
                else sc[j] = __builtin_nanf("");                      // never equal to the maximum
            }
        }
        if (__ballot(any) == 0ull) continue;
        const float m = lzw::wave_max(best);
        int chosen = -1;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            if (j < live && chosen < 0) {
                const unsigned long long hit = __ballot(sc[j] == m);
                if (hit != 0ull) chosen = j * kWave + __builtin_ctzll(hit);
            }
        }
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            if (chosen == j * kWave + lane) {
                vis[j] += 1.0f;
                vs[j] += lv[j];
                q[j] = vs[j] / fmaxf(vis[j], 1e-8f);
            }
        }
        total += 1.0f;
    }
    float sv = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int a = j * kWave + lane;
        if (a < A) { visits[root * A + a] = vis[j]; value_sum[root * A + a] = vs[j]; }
        sv += vis[j];
        sw += vs[j];
    }
    sv = wave_sum(sv);
    sw = wave_sum(sw);
    if (lane == 0) root_values[root] = sw / fmaxf(sv, 1.0f);
}

// ---- the same allocation with the two IEEE divisions of a pull replaced by exact equivalents (round 4) ------------
// A pull is instruction-issue bound -- vector AND scalar instructions count (round 5, profiles/r05_pmc_sq_bandit.md) --:
// with divisions it is ~45 vector instructions, 20 of them the two
// correctly rounded fp32 divisions  u = (c*p*sqrt_total) / (1 + visits)  and  q = value_sum / visits, ~10 more the
// correctly rounded sqrtf.  Here
//   * sqrt_total comes from a table: total == sim while pulls succeed (a pull that finds no candidate changes nothing,
//     so every later one fails too), and table[sim] = sqrtf(sim + 1) is the same correctly rounded value;
//   * x / d with d an INTEGER <= 2^17 is computed as (float)((double)x * rd[d]), rd[d] = 1.0 / d in double: the
//     product is within 2^-52 of x / d, while x / d (24-bit x, normal quotient) stays at least 2^-42 (relative) away
//     from every rounding boundary of fp32 -- x = m * d has no solution for a 25-bit odd midpoint m -- so the rounded
//     result IS the correctly rounded quotient.  The bound needs a normal quotient: waves whose priors or leaf values
//     hold non-zero magnitudes below 2^-100 (never seen; the denormal range allows exact ties) take the division path.
//   * rd[1 + visits] sits in a register pair per action; only the pulled action reloads it (PullState below).
// Bit-identical visits / value sums by construction; checked against the division kernel and the reference op (g6).
constexpr int kPuctTable = 65536;                 // pulls covered by the tables
__device__ float g_puct_sqrt[kPuctTable + 1];     // sqrtf(sim + 1); one entry more than pulls: the loops read one ahead
__device__ double g_puct_recip[kPuctTable + 3];   // 1.0 / d, d = 1 .. kPuctTable + 2 ([0] unused)
__global__ void puct_tables_kernel() {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kPuctTable + 1) g_puct_sqrt[i] = sqrtf((float)i + 1.0f);
    if (i < kPuctTable + 3) g_puct_recip[i] = i > 0 ? 1.0 / (double)i : 0.0;
}

__device__ __forceinline__ float div_by_int(float x, double rd) { return (float)((double)x * rd); }

// The pull loops get the tables as a kernel argument: the bases sit in scalar registers for the whole loop (addressed
// through the variables, the compiler re-derived the reciprocal table's address from the program counter at every pull),
// and the sqrt table is read through the constant address space, where a uniform load is a scalar load whatever else the
// kernel does (one pull ahead of its use).
typedef const __attribute__((address_space(4))) float* ConstFloats;
typedef const __attribute__((address_space(4))) char* ConstBytes;
struct PuctTables { ConstFloats sqrt_tab; const double* __restrict__ recip_tab; };

// sqrt(total + 1) of the pull being made, as a scalar one pull ahead of its use; the loop around it is a plain counted loop
// (three scalar instructions).  There is no early exit when no action has a usable score (all NaN): nothing is pulled from
// then on either way -- NaN scores stay NaN as sqrt(total) grows -- so the result is the same as the reference's `break`.
struct SqrtStream {
    ConstBytes tab;
    uint32_t off, end;
    float ahead;
    __device__ __forceinline__ SqrtStream(const PuctTables T, int sims)
        : tab((ConstBytes)T.sqrt_tab), off(0u), end(4u * (uint32_t)sims), ahead(*T.sqrt_tab) {}
    __device__ __forceinline__ bool more() const { return off < end; }
    __device__ __forceinline__ float next() {
        const float now = ahead;
        off += 4u;
        ahead = *(ConstFloats)(tab + off);                        // the table holds one entry more than pulls can be asked for
        return now;
    }
};

// One action's running numbers.  Table form: rd = 1 / (1 + visits) as a double, so u = x / (1 + visits) and, after a pull,
// q = value_sum / visits are one multiplication each (div_by_int above); the next reciprocal is fetched in the pulled lane
// straight into `rd` (addressed by 8 * visits against the table's base in scalar registers) and is first needed at the
// top of the next pull.
template <bool EXACT_DIV>
struct PullState {
    float q = 0.f;
    double rd = 1.0;
    uint32_t nv8 = 0u;                                           // 8 * visits
    __device__ __forceinline__ float score(float x, float vis) const {
        return q + (EXACT_DIV ? x / (1.0f + vis) : div_by_int(x, rd));
    }
    __device__ __forceinline__ void pull(const PuctTables T, float lv, float& vis, float& vs) {
        vis += 1.0f;
        vs += lv;
        q = EXACT_DIV ? vs / vis : div_by_int(vs, rd);
        if (!EXACT_DIV) rd = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(T.recip_tab) + nv8 + 16u);
        nv8 += 8u;
    }
};

template <int SLOTS, bool EXACT_DIV>
__device__ __forceinline__ void puct_pulls(const PuctTables T, int sims, int lane, int live, const float (&cp)[SLOTS], const float (&lv)[SLOTS],
                                           float (&vis)[SLOTS], float (&vs)[SLOTS]) {
    PullState<EXACT_DIV> st[SLOTS];
    SqrtStream sq(T, sims);
    while (sq.more()) {
        const float sqrt_total = sq.next();
        float sc[SLOTS];
        float best = -INFINITY;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            sc[j] = __builtin_nanf("");
            if (j < live) {
                sc[j] = st[j].score(cp[j] * sqrt_total, vis[j]);
                best = sc[j] > best ? sc[j] : best;
            }
        }
        const float m = lzw::wave_max_nonan(best);               // `best` is -inf or a score that won a `>`: never NaN
        int chosen = -1;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            if (j < live && chosen < 0) {
                const unsigned long long hit = __ballot(sc[j] == m);
                if (hit != 0ull) chosen = j * kWave + __builtin_ctzll(hit);
            }
        }
#pragma unroll
        for (int j = 0; j < SLOTS; ++j)
            if (chosen == j * kWave + lane) st[j].pull(T, lv[j], vis[j], vs[j]);
    }
}

// Which lane of each group of `width` lanes (16 or 32) takes the pull: the lowest one whose score is the group's maximum.
// Decided per lane from the ballot and two per-lane constants (the bits of the lane's group below the lane) -- three vector
// instructions; finding the winners with scalar bit scans per group cost ~35 scalar instructions per pull, and the scalar
// unit was what bounded the packed loops (round 5: profiles/r05_pmc_sq_bandit.md).
__device__ __forceinline__ void group_below_masks(int lane, int width, uint32_t& below_lo, uint32_t& below_hi) {
    const int first = lane & ~(width - 1);
    const unsigned long long below = ((1ull << lane) - 1ull) & ~((1ull << first) - 1ull);
    below_lo = (uint32_t)below;
    below_hi = (uint32_t)(below >> 32);
}
__device__ __forceinline__ bool first_hit_of_group(bool is_hit, unsigned long long hit, uint32_t below_lo, uint32_t below_hi) {
    return is_hit && ((((uint32_t)hit & below_lo) | ((uint32_t)(hit >> 32) & below_hi)) == 0u);
}

// Packed roots.  Two roots in one wave (root A on lanes 0-31, root B on lanes 32-63) when every valid action of both sits
// below index 32, four roots (lanes 16k .. 16k + 15 = root k) when all sit below 16: the fused root search packs its rows to
// the left, and 81 % of the positions of a game have at most 16 legal moves (g15: mean 11.7).  The same arithmetic per lane
// as puct_pulls; a pull of the pack costs about the instructions of one root's pull, so a packed quadruple takes a quarter
// of the issue slots.  Every lane gets its own group's maximum (no readlane / select per group): four cyclic rotations
// inside the rows of 16, and for halves gfx950's v_permlane16_swap on top (row 1 <-> row 0 of a copy, row 3 <-> row 2 puts
// a half's two row maxima side by side in every lane).  Two wait states separate a VALU write from a DPP / permlane
// read of the same register; the plain VALU compare that follows needs none.
__device__ __forceinline__ float row16_max(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ float half32_max(float v) {
    float t;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32 %1, %0\n\t"
                 "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\t"
                 "s_nop 1\n\tv_max_f32 %0, %0, %1"
                 : "+v"(v), "=&v"(t));
    return v;
}
// eight roots of <= 8 actions (lanes 8k .. 8k + 7 = root k): two quad permutes (lane ^ 1, lane ^ 2) and the mirror of each
// half row (lane <-> 7 - lane within 8) leave every lane with the maximum of its 8
__device__ __forceinline__ float oct8_max(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}

template <int W> __device__ __forceinline__ float group_max(float v);      // every lane: the maximum of its W-lane group
template <> __device__ __forceinline__ float group_max<32>(float v) { return half32_max(v); }
template <> __device__ __forceinline__ float group_max<16>(float v) { return row16_max(v); }
template <> __device__ __forceinline__ float group_max<8>(float v) { return oct8_max(v); }

// the pulls of 64 / W roots packed W lanes each: one ballot serves them all
template <int W, bool EXACT_DIV>
__device__ __forceinline__ void puct_pulls_packed(const PuctTables T, int sims, int lane, float cp, float lv, float& vis, float& vs) {
    uint32_t below_lo, below_hi;
    group_below_masks(lane, W, below_lo, below_hi);
    PullState<EXACT_DIV> st;
    SqrtStream sq(T, sims);
    while (sq.more()) {
        const float sc = st.score(cp * sq.next(), vis);          // NaN on lanes without a valid action
        const float m = group_max<W>(sc > -INFINITY ? sc : -INFINITY);
        const unsigned long long hit = __ballot(sc == m);
        if (first_hit_of_group(sc == m, hit, below_lo, below_hi))    // the lowest maximal lane of each group
            st.pull(T, lv, vis, vs);
    }
}

__device__ __forceinline__ bool tiny_magnitude(bool ok, float cp, float lv) {
    const float acp = fabsf(cp), alv = fabsf(lv);
    return ok && ((acp != 0.f && acp < 0x1p-100f) || (alv != 0.f && alv < 0x1p-100f));
}

// G roots (2, 4 or 8, any of them may be missing: index < 0) whose valid actions all sit below 64 / G
template <int G>
__device__ __forceinline__ void puct_group_job(const PuctTables T, const int64_t (&roots)[G], int lane, const float* __restrict__ priors,
                                               const float* __restrict__ leaf, const uint8_t* __restrict__ valid, int A,
                                               int sims, float c, float* __restrict__ visits,
                                               float* __restrict__ value_sum, float* __restrict__ root_values) {
    constexpr int W = kWave / G;                                 // lanes per root
    const int a = lane & (W - 1), k = lane / W;
    int64_t r = roots[0];
#pragma unroll
    for (int j = 1; j < G; ++j) r = k == j ? roots[j] : r;
    const bool have = r >= 0;
    const bool ok = have && a < A && valid[(have ? r : 0) * A + (a < A ? a : 0)] != 0;
    const float cp1 = ok ? c * priors[r * A + a] : __builtin_nanf("");
    const float lv1 = have && a < A ? leaf[r * A + a] : 0.f;
    float vis1 = 0.f, vs1 = 0.f;
    const bool exact = __ballot(tiny_magnitude(ok, cp1, lv1)) != 0ull;    // wave-uniform: the plain divisions
    if (exact) puct_pulls_packed<W, true>(T, sims, lane, cp1, lv1, vis1, vs1);
    else puct_pulls_packed<W, false>(T, sims, lane, cp1, lv1, vis1, vs1);
    if (have && a < A) { visits[r * A + a] = vis1; value_sum[r * A + a] = vs1; }
    if (have)
        for (int a2 = W + a; a2 < A; a2 += W) { visits[r * A + a2] = 0.f; value_sum[r * A + a2] = 0.f; }   // never visited
    float sv1 = vis1, sw1 = vs1;                                  // per-root butterfly: the same additions as wave_sum
#pragma unroll                                                    // performs for a root alone in a wave (x + 0 == x)
    for (int o = W / 2; o > 0; o >>= 1) { sv1 += __shfl_xor(sv1, o); sw1 += __shfl_xor(sw1, o); }
    if (have && a == 0) root_values[r] = sw1 / fmaxf(sv1, 1.0f);
}

template <int SLOTS>
__device__ __forceinline__ void puct_single_job(const PuctTables T, int64_t root, int lane, const float* __restrict__ priors,
                                                const float* __restrict__ leaf, const uint8_t* __restrict__ valid, int A,
                                                int sims, float c, float* __restrict__ visits,
                                                float* __restrict__ value_sum, float* __restrict__ root_values) {
    float cp[SLOTS], lv[SLOTS], vis[SLOTS], vs[SLOTS];
    int live = 0;
    bool tiny = false;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int a = j * kWave + lane;
        const bool ok = a < A && valid[root * A + (a < A ? a : 0)] != 0;
        // an action that may not be chosen carries a NaN score: never greater, never equal to the maximum
        cp[j] = ok ? c * priors[root * A + a] : __builtin_nanf("");
        lv[j] = a < A ? leaf[root * A + a] : 0.f;
        vis[j] = 0.f; vs[j] = 0.f;
        if (__ballot(ok) != 0ull) live = j + 1;
        tiny = tiny || tiny_magnitude(ok, cp[j], lv[j]);
    }
    // wave-uniform: the plain divisions for this root (two instantiations of the loop, one real branch)
    if (__ballot(tiny) != 0ull) puct_pulls<SLOTS, true>(T, sims, lane, live, cp, lv, vis, vs);
    else puct_pulls<SLOTS, false>(T, sims, lane, live, cp, lv, vis, vs);
    float sv = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int a = j * kWave + lane;
        if (a < A) { visits[root * A + a] = vis[j]; value_sum[root * A + a] = vs[j]; }
        sv += vis[j];
        sw += vs[j];
    }
    sv = wave_sum(sv);
    sw = wave_sum(sw);
    if (lane == 0) root_values[root] = sw / fmaxf(sv, 1.0f);
}

// without scratch memory (a capture in progress before any eager call): consecutive roots pair up where both fit 32 lanes
template <int SLOTS>
__global__ __launch_bounds__(kBlock) void root_puct_fast_kernel(const float* __restrict__ priors,
                                                                const float* __restrict__ leaf,
                                                                const uint8_t* __restrict__ valid, int64_t R,
                                                                int A, int sims, float c,
                                                                float* __restrict__ visits,
                                                                float* __restrict__ value_sum,
                                                                float* __restrict__ root_values, const PuctTables T) {
    const int lane = lane_id();
    const int64_t root = wave_item();
    if (root >= R) return;
    const int64_t ra = root & ~(int64_t)1, rb = ra + 1;
    bool wide = false;                                            // a valid action at index >= 32 in either row
    if (rb < R)
        for (int a = 32 + lane; a < A; a += kWave) wide = wide || valid[ra * A + a] != 0 || valid[rb * A + a] != 0;
    if (rb < R && __ballot(wide) == 0ull) {                      // the even wave of the pair works for both, the odd one leaves
        if (root != ra) return;
        const int64_t pair[2] = {ra, rb};
        puct_group_job<2>(T, pair, lane, priors, leaf, valid, A, sims, c, visits, value_sum, root_values);
        return;
    }
    puct_single_job<SLOTS>(T, root, lane, priors, leaf, valid, A, sims, c, visits, value_sum, root_values);
}

// ---- roots binned by width first (round 5): rows of <= 8 valid actions go eight to a wave (35 % of a game's positions),
// <= 16 four to a wave (46 %), <= 32 two to a wave, the rest alone -- whichever roots happen to be neighbours.  puct_bin_kernel: one wave per root finds the row's width (index of
// its last valid action + 1) and appends the root to its class list (atomics on four counters, once per ply: the list
// order varies from run to run, a root's results do not depend on its companions).  The pull kernel's wave w then takes
// octet w, quadruple w - octets, pair w - octets - quads, or a single root, from the device-side counts (no host read).
__global__ void puct_zero_counts_kernel(unsigned* counts) { if (threadIdx.x < 4) counts[threadIdx.x] = 0u; }

__global__ __launch_bounds__(kBlock) void puct_bin_kernel(const uint8_t* __restrict__ valid, int64_t R, int A,
                                                          int* __restrict__ lists, unsigned* __restrict__ counts, int64_t cap) {
    const int lane = lane_id();
    const int64_t root = wave_item();
    if (root >= R) return;
    int width = 0;
    for (int a0 = 0; a0 < A; a0 += kWave) {
        const unsigned long long b = __ballot(a0 + lane < A && valid[root * A + (a0 + lane < A ? a0 + lane : 0)] != 0);
        if (b != 0ull) width = a0 + 64 - __builtin_clzll(b);
    }
    if (lane == 0) {
        const int cls = width <= 8 ? 3 : (width <= 16 ? 0 : (width <= 32 ? 1 : 2));
        const unsigned pos = atomicAdd(&counts[cls], 1u);
        lists[(int64_t)cls * cap + pos] = (int)root;
    }
}

template <int SLOTS>
__global__ __launch_bounds__(kBlock) void root_puct_binned_kernel(const float* __restrict__ priors,
                                                                  const float* __restrict__ leaf,
                                                                  const uint8_t* __restrict__ valid, int64_t R, int A,
                                                                  int sims, float c, float* __restrict__ visits,
                                                                  float* __restrict__ value_sum,
                                                                  float* __restrict__ root_values,
                                                                  const int* __restrict__ lists,
                                                                  const unsigned* __restrict__ counts, int64_t cap,
                                                                  const PuctTables T) {
    const int lane = lane_id();
    int64_t w = wave_item();
    if (w >= R) return;
    const int64_t n0 = counts[0], n1 = counts[1], n2 = counts[2], n3 = counts[3];
    const int64_t octets = (n3 + 7) >> 3;
    if (w < octets) {
        int64_t r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = 8 * w + k < n3 ? (int64_t)lists[3 * cap + 8 * w + k] : -1;
        puct_group_job<8>(T, r, lane, priors, leaf, valid, A, sims, c, visits, value_sum, root_values);
        return;
    }
    w -= octets;
    const int64_t quads = (n0 + 3) >> 2, pairs = (n1 + 1) >> 1;
    if (w < quads) {
        int64_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = 4 * w + k < n0 ? (int64_t)lists[4 * w + k] : -1;
        puct_group_job<4>(T, r, lane, priors, leaf, valid, A, sims, c, visits, value_sum, root_values);
    } else if (w < quads + pairs) {
        const int64_t i = 2 * (w - quads);
        const int64_t r[2] = {(int64_t)lists[cap + i], i + 1 < n1 ? (int64_t)lists[cap + i + 1] : -1};
        puct_group_job<2>(T, r, lane, priors, leaf, valid, A, sims, c, visits, value_sum, root_values);
    } else if (w < quads + pairs + n2) {
        puct_single_job<SLOTS>(T, (int64_t)lists[2 * cap + (w - quads - pairs)], lane, priors, leaf, valid, A, sims, c, visits,
                               value_sum, root_values);
    }
}

// =================================================================================================
// root_finalize_from_visits (+ sampled pick): fill kernel, then one wave per root.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void finalize_fill_kernel(float* policy, int64_t n_policy, int64_t* cidx,
                                                               int32_t* ccodes, uint8_t* cvalid, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = i; j < n_policy; j += stride) policy[j] = 0.f;
    for (int64_t j = i; j < B; j += stride) { cidx[j] = -1; cvalid[j] = 0; }
    for (int64_t j = i; j < B * 4; j += stride) ccodes[j] = -1;
}

template <int SLOTS>
__global__ __launch_bounds__(kBlock) void root_finalize_kernel(
    const int64_t* __restrict__ lidx, const int4* __restrict__ codes, const uint8_t* __restrict__ valid,
    const float* __restrict__ visits, const float* __restrict__ value_sum, const int64_t* __restrict__ roots,
    int64_t R, int M, int64_t B, int T, const float* __restrict__ temps, const float* __restrict__ uniforms,
    float* __restrict__ policy, int64_t* __restrict__ cidx, int4* __restrict__ ccodes,
    uint8_t* __restrict__ cvalid, float* __restrict__ root_value) {
    const int lane = lane_id();
    const int64_t r = wave_item();
    if (r >= R) return;
    const int64_t b = roots[r];
    const float temp = fmaxf(temps[r], 1e-6f);
    const float inv_t = 1.0f / temp;
    float pol[SLOTS], vraw[SLOTS];
    bool ok[SLOTS];
    float psum = 0.f, sv = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int a = j * kWave + lane;
        const bool in = a < M;
        ok[j] = in && valid[r * M + (in ? a : 0)] != 0;
        const float v = in ? visits[r * M + a] : 0.f;
        vraw[j] = v;
        const float w = in ? value_sum[r * M + a] : 0.f;
        // module.cpp:493-495: pow(clamp_min(visits,1e-8), 1/T) * mask, normalised (sum clamp 1e-8)
        pol[j] = ok[j] ? powf(fmaxf(v, 1e-8f), inv_t) : 0.f;
        psum += pol[j];
        sv += v;
        sw += w;
    }
    {   // a row without any valid action (a terminal root in the padded layout of the fused search) keeps the fill
        // kernel's defaults: zero policy, index -1, not valid
        bool any = false;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) any = any || ok[j];
        if (__ballot(any) == 0ull) return;
    }
    psum = fmaxf(wave_sum(psum), 1e-8f);
    sv = wave_sum(sv);
    sw = wave_sum(sw);
    // argmax of the normalised policy, first index wins (module.cpp:501)
    unsigned long long key = 0ull;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        pol[j] = pol[j] / psum;
        const int a = j * kWave + lane;
        if (a < M && pol[j] == pol[j]) {
            const unsigned long long k = ((unsigned long long)float_order(pol[j]) << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (uint32_t)a);
            key = k > key ? k : key;
        }
    }
    key = wave_max_u64(key);
    int pick_local = key ? (int)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) : 0;

    if (uniforms != nullptr && M > 1) {
        // mcts_gpu.py:853-898: softmax(log(visits)/T) over legal actions, then inverse-CDF sampling
        float lg[SLOTS];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            float v = vraw[j];
            if (!(v == v) || isinf(v)) v = 0.f;
            lg[j] = ok[j] ? logf(fmaxf(v, 1e-8f)) * inv_t : -INFINITY;
            mx = fmaxf(mx, lg[j]);
        }
        mx = wave_max(mx);
        if (!isfinite(mx)) mx = 0.f;
        float ex[SLOTS];
        float esum = 0.f;
        int nvalid = 0;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            ex[j] = ok[j] ? expf(lg[j] - mx) : 0.f;
            if (!(ex[j] == ex[j]) || isinf(ex[j])) ex[j] = 0.f;
            esum += ex[j];
            nvalid += ok[j] ? 1 : 0;
        }
        esum = wave_sum(esum);
        nvalid = (int)wave_sum((float)nvalid);
        if (!(esum > 0.f) || !isfinite(esum)) {          // fallback: uniform over legal
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) ex[j] = ok[j] ? 1.0f : 0.f;
            esum = (float)(nvalid > 0 ? nvalid : 1);
        }
        // inclusive prefix over action order: slot-major (j), then lane
        const float target = uniforms[r] * esum;
        float run = 0.f;
        int chosen = -1, last_valid = -1;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            float incl = ex[j];
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const float t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const float cum = run + incl;
            const bool hit = ok[j] && ex[j] > 0.f && cum > target;
            const uint64_t hb = __ballot(hit);
            if (chosen < 0 && hb) chosen = j * kWave + (__ffsll((unsigned long long)hb) - 1);
            const uint64_t vb = __ballot(ok[j] && ex[j] > 0.f);
            if (vb) last_valid = j * kWave + (63 - __clzll((unsigned long long)vb));
            run += __shfl(incl, kWave - 1);
        }
        if (chosen < 0) chosen = last_valid;             // rounding at the top end
        if (chosen >= 0) pick_local = chosen;
    }

    // dense scatter (policy_dense_valid.scatter_add_ + index_copy_), chosen index / code
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int a = j * kWave + lane;
        if (a < M && ok[j]) {
            const int64_t col = lidx[r * M + a];
            if (col >= 0 && col < T) atomicAdd(&policy[b * T + col], pol[j]);
        }
    }
    if (lane == 0) {
        cidx[b] = lidx[r * M + pick_local];
        ccodes[b] = codes[r * M + pick_local];
        cvalid[b] = 1;
        root_value[r] = sw / fmaxf(sv, 1.0f);
    }
}

// =================================================================================================
// self_play_step_inplace: one lane per active game.
// =================================================================================================
__device__ __forceinline__ float soft_value(uint64_t black, uint64_t white, float k) {
    const float delta = (float)(popc(black) - popc(white)) / 18.0f;
    return tanhf(delta * k);
}

__global__ __launch_bounds__(kBlock) void self_play_step_kernel(
    LzStateSoA s, int64_t B, int64_t* __restrict__ plies, uint8_t* __restrict__ done,
    const int64_t* __restrict__ active, int64_t n_active, const int4* __restrict__ codes,
    const uint8_t* __restrict__ terminal, const uint8_t* __restrict__ cvalid, int64_t max_plies, float k,
    int32_t* __restrict__ fin_kind, float* __restrict__ result, float* __restrict__ soft) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_active) return;
    const int64_t slot = active[i];
    if (slot < 0 || slot >= B) { fin_kind[i] = 0; result[i] = 0.f; soft[i] = 0.f; return; }
    State st = load_state(s, slot);
    const bool term = terminal[i] != 0;
    if (term || cvalid[i] == 0) {                      // module.cpp:724-741
        done[slot] = 1;
        fin_kind[i] = 1;
        result[i] = term ? -(float)s.current_player[slot] : 0.f;
        soft[i] = soft_value(st.black, st.white, k);
        return;
    }
    const int4 code = codes[i];
    apply(st, code.x, code.y, code.z);
    store_state(s, slot, st);
    const int64_t np = plies[slot] + 1;
    plies[slot] = np;
    int winner = 0;                                     // module.cpp:822-836
    const bool post = st.phase == kMovement || st.phase == kCaptureSelection || st.phase == kCounterRemoval;
    if (post && popc(st.black) < kLoseThreshold) winner = -1;
    if (post && popc(st.white) < kLoseThreshold) winner = 1;
    const bool draw = st.move_count >= kMaxMoveCount || st.msc >= kNoCaptureLimit;
    const bool cap = np >= max_plies;
    if (winner != 0 || draw || cap) {
        done[slot] = 1;
        fin_kind[i] = 2;
        result[i] = (float)winner;
        soft[i] = soft_value(st.black, st.white, k);
    } else {
        fin_kind[i] = 0;
        result[i] = 0.f;
        soft[i] = 0.f;
    }
}

// =================================================================================================
// finalize_trajectory_inplace: one wave per finished game, lanes stride over its recorded steps.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void finalize_trajectory_kernel(
    float* __restrict__ value_t, float* __restrict__ soft_t, const int8_t* __restrict__ signs,
    const int64_t* __restrict__ step_index, const int64_t* __restrict__ step_counts, int64_t G, int64_t Tmax,
    const int64_t* __restrict__ slots, const float* __restrict__ result, const float* __restrict__ softv,
    int64_t F, uint8_t* __restrict__ keep, int64_t* __restrict__ final_counts,
    unsigned long long* __restrict__ counts_out) {
    const int lane = lane_id();
    const int64_t f = wave_item();
    if (f >= F) return;
    const int64_t g = slots[f];
    int64_t n = (g >= 0 && g < G) ? step_counts[g] : 0;
    if (n > Tmax) n = Tmax;
    const float res = result[f], sft = softv[f];
    if (lane == 0) {
        keep[f] = n > 0 ? 1 : 0;
        final_counts[f] = n;
        if (n > 0) atomicAdd(&counts_out[res > 0.f ? 0 : (res < 0.f ? 1 : 2)], 1ull);
    }
    for (int64_t j = lane; j < n; j += kWave) {
        const int64_t idx = step_index[g * Tmax + j];
        const float sg = (float)signs[idx];
        value_t[idx] = sg * res;
        soft_t[idx] = sg * sft;
    }
}

// =================================================================================================
// Fused per-ply tail of the wave loop for a FIXED wave of G slots (finished slots stay in the batch and are
// masked by `done`): trajectory rows are assigned and written on the device, the move is applied and finished
// games are finalised (and optionally re-seated) without a host round trip.
// =================================================================================================
constexpr int kRowsBlock = 1024;
// rows[g] = *cursor + (number of live slots below g), -1 for finished slots; step_index / step_counts updated.
__global__ __launch_bounds__(kRowsBlock) void wave_rows_kernel(
    const uint8_t* __restrict__ done, int64_t G, int64_t* __restrict__ cursor, int64_t capacity, int64_t Tmax,
    int64_t* __restrict__ step_index, int64_t* __restrict__ step_counts, int64_t* __restrict__ rows,
    int32_t* __restrict__ overflow) {
    __shared__ int wave_total[kRowsBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int64_t per = (G + kRowsBlock - 1) / kRowsBlock;
    const int64_t lo = tid * per, hi = (lo + per < G) ? lo + per : G;
    int cnt = 0;
    for (int64_t j = lo; j < hi; ++j) cnt += done[j] == 0 ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
    }
    if (lane == kWave - 1) wave_total[w] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < kRowsBlock / kWave; ++i) { if (i < w) before += wave_total[i]; total += wave_total[i]; }
    int64_t row = *cursor + before + incl - cnt;
    __syncthreads();                                    // every thread has read the cursor
    for (int64_t j = lo; j < hi; ++j) {
        if (done[j] != 0) { rows[j] = -1; continue; }
        const int64_t n = step_counts[j];
        if (row >= capacity || n >= Tmax) { rows[j] = -1; atomicAdd(overflow, 1); }
        else { rows[j] = row; step_index[j * Tmax + n] = row; step_counts[j] = n + 1; }
        ++row;
    }
    if (tid == 0) *cursor += total;
}

// Slot-major live arena (the finished-row log, lz_wave_log_finished): the sample of slot g's step n lives in row
// g * Tmax + n until the game ends, so there is nothing to scan and no step_index matrix.
__global__ __launch_bounds__(kBlock) void wave_rows_slot_kernel(
    const uint8_t* __restrict__ done, int64_t G, int64_t Tmax, int64_t* __restrict__ step_counts,
    int64_t* __restrict__ rows, int32_t* __restrict__ overflow) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= G) return;
    if (done[g] != 0) { rows[g] = -1; return; }
    const int64_t n = step_counts[g];
    if (n >= Tmax) { rows[g] = -1; atomicAdd(overflow, 1); return; }
    rows[g] = g * Tmax + n;
    step_counts[g] = n + 1;
}

// Start the next games in finished slots, in ascending slot order while the budget lasts (deterministic): slot g
// restarts from the empty board as game *next_game + rank(g).  One workgroup, same scan as wave_rows_kernel.
// `logged_only`: a finished slot whose rows have not left for the finished-row log yet (step_counts > 0) waits.
__global__ __launch_bounds__(kRowsBlock) void wave_reseat_kernel(
    LzStateSoA s, int64_t G, uint8_t* __restrict__ done, int64_t* __restrict__ plies,
    int64_t* __restrict__ step_counts, int64_t* __restrict__ budget, int64_t* __restrict__ next_game,
    int64_t* __restrict__ slot_game, uint8_t* __restrict__ reseated, int logged_only) {
    __shared__ int wave_total[kRowsBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int64_t left = *budget;
    if (left <= 0) return;                              // uniform: the common case once every game has started
    const int64_t per = (G + kRowsBlock - 1) / kRowsBlock;
    const int64_t lo = tid * per, hi = (lo + per < G) ? lo + per : G;
    int cnt = 0;
    for (int64_t j = lo; j < hi; ++j) cnt += (done[j] != 0 && !(logged_only && step_counts[j] > 0)) ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
    }
    if (lane == kWave - 1) wave_total[w] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < kRowsBlock / kWave; ++i) { if (i < w) before += wave_total[i]; total += wave_total[i]; }
    const int64_t first = *next_game;
    int64_t rank = before + incl - cnt;
    __syncthreads();                                    // every thread has read budget / next_game
    for (int64_t j = lo; j < hi; ++j) {
        if (done[j] == 0 || (logged_only && step_counts[j] > 0)) continue;
        if (rank < left) {
            State fresh{};
            fresh.phase = kPlacement;
            fresh.player = 1;
            store_state(s, j, fresh);
            plies[j] = 0;
            step_counts[j] = 0;
            done[j] = 0;
            slot_game[j] = first + rank;
            if (reseated) reseated[j] = 1;
        }
        ++rank;
    }
    if (tid == 0) {
        const int64_t used = total < left ? total : left;
        *budget = left - used;
        *next_game = first + used;
    }
}

// one wave per slot: copy the ply's sample into arena row rows[g] (value targets start as NaN)
__global__ __launch_bounds__(kBlock) void wave_record_kernel(
    const int64_t* __restrict__ rows, int64_t G, const float* __restrict__ model_input,
    const uint8_t* __restrict__ legal, const float* __restrict__ policy, const int64_t* __restrict__ player, int T,
    float* __restrict__ a_state, uint8_t* __restrict__ a_legal, float* __restrict__ a_policy,
    float* __restrict__ a_value, float* __restrict__ a_soft, int8_t* __restrict__ a_sign) {
    const int lane = lane_id();
    const int64_t g = wave_item();
    if (g >= G) return;
    const int64_t row = rows[g];
    if (row < 0) return;
    constexpr int kIn = 11 * 36;
    for (int j = lane; j < kIn; j += kWave) a_state[row * kIn + j] = model_input[g * kIn + j];
    for (int j = lane; j < T; j += kWave) {
        a_legal[row * T + j] = legal[g * T + j];
        a_policy[row * T + j] = policy[g * T + j];
    }
    if (lane == 0) {
        a_value[row] = __builtin_nanf("");
        a_soft[row] = __builtin_nanf("");
        a_sign[row] = player[g] >= 0 ? 1 : -1;
    }
}

// one wave per slot: lane 0 plays the move (same rules as self_play_step_kernel), then the wave finalises a
// finished game's rows (finalize_trajectory_kernel) and books it; `reseat` puts a fresh game into the slot.
__global__ __launch_bounds__(kBlock) void wave_step_finish_kernel(
    LzStateSoA s, int64_t G, int64_t* __restrict__ plies, uint8_t* __restrict__ done, const int4* __restrict__ codes,
    const uint8_t* __restrict__ terminal, const uint8_t* __restrict__ cvalid, int64_t max_plies, float k,
    float* __restrict__ value_t, float* __restrict__ soft_t, const int8_t* __restrict__ signs,
    const int64_t* __restrict__ step_index, int64_t* __restrict__ step_counts, int64_t Tmax,
    unsigned long long* __restrict__ outcome, unsigned long long* __restrict__ delta_hist,
    int64_t* __restrict__ lengths, const int64_t* __restrict__ slot_game, unsigned long long* __restrict__ finished,
    uint8_t* __restrict__ reseated, int reseat) {
    const int lane = lane_id();
    const int64_t g = wave_item();
    if (g >= G) return;
    if (done[g] != 0) return;
    int fin = 0, delta = 0;
    float res = 0.f, sft = 0.f;
    if (lane == 0) {
        State st = load_state(s, g);
        const bool term = terminal[g] != 0;
        if (term || cvalid[g] == 0) {
            fin = 1;
            res = term ? -(float)s.current_player[g] : 0.f;
        } else {
            const int4 code = codes[g];
            apply(st, code.x, code.y, code.z);
            store_state(s, g, st);
            const int64_t np = plies[g] + 1;
            plies[g] = np;
            int winner = 0;
            const bool post = st.phase == kMovement || st.phase == kCaptureSelection || st.phase == kCounterRemoval;
            if (post && popc(st.black) < kLoseThreshold) winner = -1;
            if (post && popc(st.white) < kLoseThreshold) winner = 1;
            const bool draw = st.move_count >= kMaxMoveCount || st.msc >= kNoCaptureLimit;
            if (winner != 0 || draw || np >= max_plies) { fin = 2; res = (float)winner; }
        }
        if (fin) { sft = soft_value(st.black, st.white, k); delta = popc(st.black) - popc(st.white); }
    }
    fin = __shfl(fin, 0, kWave);
    if (fin == 0) return;
    res = __shfl(res, 0, kWave);
    sft = __shfl(sft, 0, kWave);
    int64_t n = step_counts[g];
    if (n > Tmax) n = Tmax;
    for (int64_t j = lane; j < n; j += kWave) {
        const int64_t idx = step_index ? step_index[g * Tmax + j] : g * Tmax + j;
        const float sg = (float)signs[idx];
        value_t[idx] = sg * res;
        soft_t[idx] = sg * sft;
    }
    if (lane == 0) {
        if (n > 0) {
            atomicAdd(&outcome[res > 0.f ? 0 : (res < 0.f ? 1 : 2)], 1ull);
            if (lengths) lengths[slot_game ? slot_game[g] : g] = n;
        }
        if (delta_hist) atomicAdd(&delta_hist[delta < -18 ? 0 : (delta > 18 ? 36 : delta + 18)], 1ull);
        if (finished) atomicAdd(finished, 1ull);
        if (reseat) {
            State fresh{};
            fresh.phase = kPlacement;
            fresh.player = 1;
            store_state(s, g, fresh);
            plies[g] = 0;
            step_counts[g] = 0;
            if (reseated) reseated[g] = 1;
        } else {
            done[g] = 1;
        }
    }
}

// ---- finished-row log: the rows of a game leave the slot-major live arena for a game-major log the moment the game
// has ended, so that a consumer can take finished samples away WHILE the wave goes on playing (the worker streams
// them to the host; v1/python/self_play_worker.py:430-546 only sees its rows after a whole wave has drained).
// Plan (one workgroup, ordered scan -> deterministic log order): the finished slots that still hold rows, in
// ascending slot order, get log rows [cursor + prefix, +n) while they fit `capacity`; the others wait (back-pressure:
// wave_reseat_kernel does not re-seat them, the host switches to an empty log within a few plies).
// log_state int64[4] = {rows in the log, games in the log, games waiting, rows waiting}.
__global__ __launch_bounds__(kRowsBlock) void wave_log_plan_kernel(
    const uint8_t* __restrict__ done, const int64_t* __restrict__ step_counts, int64_t G, int64_t capacity,
    int64_t* __restrict__ log_state, int64_t* __restrict__ log_base) {
    __shared__ long long wave_rows[kRowsBlock / kWave];
    __shared__ int fit_games[kRowsBlock / kWave], wait_games[kRowsBlock / kWave];
    __shared__ long long fit_rows[kRowsBlock / kWave], wait_rows[kRowsBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int64_t per = (G + kRowsBlock - 1) / kRowsBlock;
    const int64_t lo = tid * per, hi = (lo + per < G) ? lo + per : G;
    long long cnt = 0;
    for (int64_t j = lo; j < hi; ++j) cnt += (done[j] != 0 && step_counts[j] > 0) ? step_counts[j] : 0;
    long long incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const long long v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
    }
    if (lane == kWave - 1) wave_rows[w] = incl;
    __syncthreads();
    long long before = 0;
    for (int i = 0; i < w; ++i) before += wave_rows[i];
    const long long cursor = log_state[0];
    long long pre = before + incl - cnt;                // rows of the waiting slots below this thread's range
    int fg = 0, wg = 0;
    long long fr = 0, wr = 0;
    for (int64_t j = lo; j < hi; ++j) {
        const long long n = (done[j] != 0 && step_counts[j] > 0) ? step_counts[j] : 0;
        if (n == 0) { log_base[j] = -1; continue; }
        if (cursor + pre + n <= capacity) { log_base[j] = cursor + pre; ++fg; fr += n; }
        else { log_base[j] = -1; ++wg; wr += n; }
        pre += n;
    }
    // prefixes are monotone, so the slots that fit are a prefix of the waiting list: their rows are contiguous
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        fg += __shfl_xor(fg, d, kWave); wg += __shfl_xor(wg, d, kWave);
        fr += __shfl_xor(fr, d, kWave); wr += __shfl_xor(wr, d, kWave);
    }
    if (lane == 0) { fit_games[w] = fg; wait_games[w] = wg; fit_rows[w] = fr; wait_rows[w] = wr; }
    __syncthreads();                                    // also: every thread has read the cursor
    if (tid == 0) {
        long long a = 0, b = 0, c = 0, d2 = 0;
        for (int i = 0; i < kRowsBlock / kWave; ++i) { a += fit_rows[i]; b += fit_games[i]; c += wait_games[i]; d2 += wait_rows[i]; }
        log_state[0] = cursor + a;
        log_state[1] += b;
        log_state[2] = c;
        log_state[3] = d2;
    }
}

// one wave per slot: the n rows of slot g (contiguous in the slot-major arena) -> log rows [log_base[g], +n)
__global__ __launch_bounds__(kBlock) void wave_log_copy_kernel(
    const int64_t* __restrict__ log_base, int64_t* __restrict__ step_counts, int64_t G, int64_t Tmax, int T,
    const float* __restrict__ a_state, const uint8_t* __restrict__ a_legal, const float* __restrict__ a_policy,
    const float* __restrict__ a_value, const float* __restrict__ a_soft, float* __restrict__ l_state,
    uint8_t* __restrict__ l_legal, float* __restrict__ l_policy, float* __restrict__ l_value,
    float* __restrict__ l_soft) {
    const int lane = lane_id();
    const int64_t g = wave_item();
    if (g >= G) return;
    const int64_t base = log_base[g];
    if (base < 0) return;
    const int64_t n = step_counts[g], src = g * Tmax;
    constexpr int kIn = 11 * 36;                         // 1 584 B: float4 copies (row starts are 16-byte multiples)
    {
        const float4* in = reinterpret_cast<const float4*>(a_state + src * kIn);
        float4* out = reinterpret_cast<float4*>(l_state + base * kIn);
        for (int64_t j = lane; j < n * (kIn / 4); j += kWave) out[j] = in[j];
    }
    if ((T & 3) == 0) {
        const float4* in = reinterpret_cast<const float4*>(a_policy + src * T);
        float4* out = reinterpret_cast<float4*>(l_policy + base * T);
        for (int64_t j = lane; j < n * (T / 4); j += kWave) out[j] = in[j];
        const uint32_t* li = reinterpret_cast<const uint32_t*>(a_legal + src * T);
        uint32_t* lo = reinterpret_cast<uint32_t*>(l_legal + base * T);
        for (int64_t j = lane; j < n * (T / 4); j += kWave) lo[j] = li[j];
    } else {
        for (int64_t j = lane; j < n * T; j += kWave) {
            l_policy[base * T + j] = a_policy[src * T + j];
            l_legal[base * T + j] = a_legal[src * T + j];
        }
    }
    for (int64_t j = lane; j < n; j += kWave) {
        l_value[base + j] = a_value[src + j];
        l_soft[base + j] = a_soft[src + j];
    }
    if (lane == 0) step_counts[g] = 0;                  // logged: the slot may start its next game
}

inline unsigned grid_waves(int64_t items) { return (unsigned)((items + kWavesPerBlock - 1) / kWavesPerBlock); }
inline unsigned grid_threads(int64_t items) { return (unsigned)((items + kBlock - 1) / kBlock); }

inline bool soa_ok(const LzStateSoA* s) {
    return s && s->board && s->marks_black && s->marks_white && s->phase && s->current_player &&
           s->pending_marks_required && s->pending_marks_remaining && s->pending_captures_required &&
           s->pending_captures_remaining && s->forced_removals_done && s->move_count && s->moves_since_capture;
}
inline bool soa_aligned(const LzStateSoA* s) {
    return aligned(s->board, 4) && aligned(s->marks_black, 4) && aligned(s->marks_white, 4);
}

}  // namespace

// Opening plies (self_play_gpu_runner.py `opening_random_moves`; mcts_gpu.py:1425-1447): the move of a flagged root is a
// uniform pick among its valid actions -- the k-th valid slot of its packed row in ascending order, k = min(floor(u n),
// n - 1) -- whatever the search chose; its policy target stays the search's.  One wave per row.
__global__ __launch_bounds__(kBlock) void root_force_uniform_kernel(
    const int64_t* __restrict__ lidx, const int4* __restrict__ codes, const uint8_t* __restrict__ valid,
    const int64_t* __restrict__ roots, int64_t R, int M, const uint8_t* __restrict__ force, const float* __restrict__ uniforms,
    int64_t* __restrict__ cidx, int4* __restrict__ ccodes, uint8_t* __restrict__ cvalid) {
    const int lane = lane_id();
    const int64_t r = wave_item();
    if (r >= R) return;
    const int64_t b = roots ? roots[r] : r;
    if (force[b] == 0) return;
    int n = 0;
    for (int a0 = 0; a0 < M; a0 += kWave)
        n += __popcll(__ballot(a0 + lane < M && valid[r * M + (a0 + lane < M ? a0 + lane : 0)] != 0));
    if (n == 0) return;                                          // no legal action: the search's (invalid) pick stands
    int k = (int)(uniforms[r] * (float)n);
    k = k < 0 ? 0 : (k >= n ? n - 1 : k);
    for (int a0 = 0; a0 < M; a0 += kWave) {
        const bool v = a0 + lane < M && valid[r * M + (a0 + lane < M ? a0 + lane : 0)] != 0;
        const unsigned long long bal = __ballot(v);
        const int here = __popcll(bal);
        if (k < here) {
            if (v && __popcll(bal & ((1ull << lane) - 1ull)) == k) {
                cidx[b] = lidx[r * M + a0 + lane];
                ccodes[b] = codes[r * M + a0 + lane];
                cvalid[b] = 1;
            }
            return;
        }
        k -= here;
    }
}

// Scratch memory of the binned bandit: three lists of `cap` root indices + the class counts, one block per (device,
// stream) so that two streams never share lists.  Allocated by the first eager call that needs it; a call on a capturing
// stream only uses what exists already (hipMalloc is not capturable).  A block that is too small is replaced, the old one is
// kept alive: captured graphs may still point at it.
constexpr int kPuctLists = 4;                                  // class lists: <= 16, <= 32, wider, <= 8 (each `cap` entries)
struct PuctScratch { int* lists; unsigned* counts; int64_t cap; };
static bool puct_scratch(int device, hipStream_t st, int64_t R, PuctScratch* out) {
    struct Entry { int device; hipStream_t st; PuctScratch s; };
    static std::mutex mu;
    static std::vector<Entry> entries;                            // newest last; superseded blocks stay (never freed)
    std::lock_guard<std::mutex> lk(mu);
    for (auto it = entries.rbegin(); it != entries.rend(); ++it)
        if (it->device == device && it->st == st && it->s.cap >= R) { *out = it->s; return true; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    PuctScratch s{};
    s.cap = R < 1024 ? 1024 : R;
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)s.cap * kPuctLists * sizeof(int) + 4 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
    s.lists = static_cast<int*>(p);
    s.counts = reinterpret_cast<unsigned*>(s.lists + kPuctLists * s.cap);
    entries.push_back({device, st, s});
    *out = s;
    return true;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* lz_version(void) { return "liuzhou-hip 0.1 (gfx950)"; }

const char* lz_status_string(int status) {
    switch (status) {
        case LZ_OK: return "ok";
        case LZ_ERR_ARG: return "invalid argument";
        case LZ_ERR_UNSUPPORTED: return "unsupported dimensions";
        case LZ_ERR_LAUNCH: return "kernel launch failed";
        case LZ_ERR_ALIGN: return "misaligned pointer";
        case LZ_ERR_ILLEGAL: return "illegal action";
        default: return "unknown status";
    }
}

int lz_encode_actions_fast(const LzStateSoA* s, int64_t B, int64_t pd, int64_t md, int64_t sd, int64_t ad,
                           uint8_t* mask, int32_t* meta, void* stream) {
    if (B < 0 || ad < 0) return LZ_ERR_ARG;
    if (pd != 36 || md != 144 || sd != 36 || ad > 40) return LZ_ERR_UNSUPPORTED;
    if (B == 0) return LZ_OK;                          /* empty batch: pointers may be null */
    if (!soa_ok(s) || !mask || !meta) return LZ_ERR_ARG;
    if (!aligned(meta, 16)) return LZ_ERR_ALIGN;
    const int T = (int)(pd + md + sd + ad);
    if ((T % 4) == 0 && aligned(mask, 4))
        hipLaunchKernelGGL(encode_actions_kernel<true>, dim3(grid_waves(B)), dim3(kBlock), 0, as_stream(stream), *s, B, T, mask, meta);
    else
        hipLaunchKernelGGL(encode_actions_kernel<false>, dim3(grid_waves(B)), dim3(kBlock), 0, as_stream(stream), *s, B, T, mask, meta);
    return launch_status();
}

int lz_batch_apply_moves(const LzStateSoA* s, int64_t B, const int32_t* codes, const int64_t* parents,
                         int64_t N, const LzStateSoA* out, void* stream) {
    if (B < 0 || N < 0) return LZ_ERR_ARG;
    if (N == 0) return LZ_OK;
    if (!s || !out || !soa_ok(s) || !soa_ok(out) || !codes || !parents) return LZ_ERR_ARG;
    if (!soa_aligned(s) || !soa_aligned(out) || !aligned(codes, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(apply_moves_kernel, dim3(grid_threads(N)), dim3(kBlock), 0, as_stream(stream), *s, B,
                       reinterpret_cast<const int4*>(codes), parents, N, *out);
    return launch_status();
}

int lz_batch_apply_moves_inplace(const LzStateSoA* s, int64_t B, const int32_t* codes, const int64_t* slots,
                                 int64_t N, void* stream) {
    if (B < 0 || N < 0) return LZ_ERR_ARG;
    if (N == 0) return LZ_OK;
    if (!soa_ok(s) || !codes || !slots) return LZ_ERR_ARG;
    if (!soa_aligned(s) || !aligned(codes, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(apply_moves_inplace_kernel, dim3(grid_threads(N)), dim3(kBlock), 0, as_stream(stream), *s, B,
                       reinterpret_cast<const int4*>(codes), slots, N);
    return launch_status();
}

int lz_states_to_model_input(const int8_t* board, const uint8_t* mb, const uint8_t* mw, const int64_t* phase,
                             const int64_t* player, int64_t B, float* out, void* stream) {
    if (B < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!board || !mb || !mw || !phase || !player || !out) return LZ_ERR_ARG;
    if (!aligned(out, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(model_input_kernel, dim3(grid_waves(B)), dim3(kBlock), 0, as_stream(stream), board, mb, mw,
                       phase, player, B, out);
    return launch_status();
}

int lz_project_policy_logits_fast(const float* lp1, const float* lp2, const float* lpmc, const uint8_t* mask,
                                  int64_t B, int64_t pd, int64_t md, int64_t sd, int64_t ad, float* probs,
                                  float* masked_logits, void* stream) {
    if (B < 0 || ad < 0) return LZ_ERR_ARG;
    if (pd != 36 || md != 144 || sd != 36 || ad > 40) return LZ_ERR_UNSUPPORTED;
    if (B == 0) return LZ_OK;
    if (!lp1 || !lp2 || !lpmc || !mask || !probs || !masked_logits) return LZ_ERR_ARG;
    const int T = (int)(pd + md + sd + ad);
    hipLaunchKernelGGL(project_policy_kernel, dim3(grid_waves(B)), dim3(kBlock), 0, as_stream(stream), lp1, lp2, lpmc,
                       mask, B, T, probs, masked_logits);
    return launch_status();
}

int lz_root_pack_rows(const uint8_t* mask, const float* probs, const int32_t* meta, int64_t B, int64_t T,
                      int64_t cap, int32_t* counts, int32_t* legal_index, float* priors, int32_t* codes,
                      void* stream) {
    if (B < 0 || T <= 0 || cap <= 0) return LZ_ERR_ARG;
    if (T > 4096 || cap > 4096) return LZ_ERR_UNSUPPORTED;
    if (B == 0) return LZ_OK;
    if (!mask || !probs || !meta || !counts || !legal_index || !priors || !codes) return LZ_ERR_ARG;
    if (!aligned(meta, 16) || !aligned(codes, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(root_pack_kernel, dim3(grid_waves(B)), dim3(kBlock), 0, as_stream(stream), mask, probs,
                       reinterpret_cast<const int4*>(meta), B, (int)T, (int)cap, counts, legal_index, priors,
                       reinterpret_cast<int4*>(codes));
    return launch_status();
}

int lz_root_pack_plan(const int32_t* counts, int64_t B, int32_t* rank, int64_t* child_off, int64_t* sizes,
                      void* stream) {
    if (B < 0) return LZ_ERR_ARG;
    if (!sizes) return LZ_ERR_ARG;
    if (B > 0 && (!counts || !rank || !child_off)) return LZ_ERR_ARG;
    hipLaunchKernelGGL(root_pack_plan_kernel, dim3(1), dim3(kPlanBlock), 0, as_stream(stream), counts, B, rank, child_off,
                       sizes);
    return launch_status();
}

int lz_root_pack_fill(const int32_t* counts, const int32_t* legal_index, const float* priors, const int32_t* codes,
                      const int32_t* rank, const int64_t* child_off, int64_t B, int64_t cap, int64_t R, int64_t M,
                      int64_t N, uint8_t* terminal_mask, int64_t* valid_root_indices, int64_t* counts_out,
                      uint8_t* valid_mask, int64_t* legal_index_mat, float* priors_mat, int32_t* action_code_mat,
                      int64_t* pack_flat_idx, int32_t* action_codes_all, int64_t* parent_indices_all, void* stream) {
    if (B < 0 || cap < 1 || R < 0 || M < 0 || N < 0 || R > B || M > cap) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!counts || !legal_index || !priors || !codes || !rank || !child_off || !terminal_mask) return LZ_ERR_ARG;
    if (R > 0 && (!valid_root_indices || !counts_out)) return LZ_ERR_ARG;
    if (R * M > 0 && (!valid_mask || !legal_index_mat || !priors_mat || !action_code_mat)) return LZ_ERR_ARG;
    if (N > 0 && (!pack_flat_idx || !action_codes_all || !parent_indices_all)) return LZ_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(codes) | reinterpret_cast<uintptr_t>(action_code_mat) |
         reinterpret_cast<uintptr_t>(action_codes_all)) & 15)
        return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(root_pack_fill_kernel, dim3(grid_waves(B)), dim3(kBlock), 0, as_stream(stream), counts, legal_index,
                       priors, reinterpret_cast<const int4*>(codes), rank, child_off, B, (int)cap, (int)M, terminal_mask,
                       valid_root_indices, counts_out, valid_mask, legal_index_mat, priors_mat,
                       reinterpret_cast<int4*>(action_code_mat), pack_flat_idx, reinterpret_cast<int4*>(action_codes_all),
                       parent_indices_all);
    return launch_status();
}

static int64_t puct_workspace_bytes(int64_t num_roots) {
    return (num_roots < 0 ? 0 : num_roots) * kPuctLists * (int64_t)sizeof(int) + 4 * (int64_t)sizeof(unsigned);
}

static int root_puct_impl(const float* priors, const float* leaf, const uint8_t* valid, int64_t R, int64_t A, int64_t sims,
                          float c, float* visits, float* value_sum, float* root_values, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    if (R < 0 || A < 0 || sims <= 0) return LZ_ERR_ARG;
    if (A > 256) return LZ_ERR_UNSUPPORTED;
    if (R == 0 || A == 0) return LZ_OK;
    if (!priors || !leaf || !valid || !visits || !value_sum || !root_values) return LZ_ERR_ARG;
    const dim3 grid(grid_waves(R)), block(kBlock);
    hipStream_t st = as_stream(stream);
    // table-driven pulls (bit-identical, about half the instructions) for budgets the tables cover;
    // LZ_ROOT_PUCT_DIV=1 forces the division kernel (tests compare the two)
    const char* force = getenv("LZ_ROOT_PUCT_DIV");
    bool use_tables = sims <= kPuctTable && !(force && force[0] == '1');
    int device = 0;
    if (use_tables) {
        // The tables are filled ONCE per device, synchronously: the flag is only set after the fill has completed and
        // been checked, so no launch on any stream can read a partial table.  A stream that is being
        // captured cannot be synchronised: such a call takes the division kernel until an eager call has filled the
        // tables (FusedRootSearch makes one before it captures).
        static std::mutex tables_mu;
        static bool tables_ready[64] = {};
        if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return LZ_ERR_LAUNCH;
        std::lock_guard<std::mutex> lk(tables_mu);
        if (!tables_ready[device]) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) return LZ_ERR_LAUNCH;
            if (cs == hipStreamCaptureStatusNone) {
                hipLaunchKernelGGL(puct_tables_kernel, dim3((kPuctTable + 3 + 255) / 256), dim3(256), 0, st);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return LZ_ERR_LAUNCH;
                tables_ready[device] = true;
            } else {
                use_tables = false;
            }
        }
    }
    if (use_tables) {
        PuctTables tables{};
        {
            void *ps = nullptr, *pr = nullptr;
            if (hipGetSymbolAddress(&ps, HIP_SYMBOL(g_puct_sqrt)) != hipSuccess ||
                hipGetSymbolAddress(&pr, HIP_SYMBOL(g_puct_recip)) != hipSuccess)
                return LZ_ERR_LAUNCH;
            tables.sqrt_tab = (ConstFloats)reinterpret_cast<uintptr_t>(ps);
            tables.recip_tab = static_cast<const double*>(pr);
        }
        // binned by width when this stream has scratch memory for the lists (allocated once per device and stream by an
        // eager call; never during a capture -- FusedRootSearch warms up eagerly before it captures); LZ_ROOT_PUCT_BIN=0
        // keeps neighbours-pair-up (tests compare the two)
        const char* nobin = getenv("LZ_ROOT_PUCT_BIN");
        PuctScratch sc{};
        bool have = false;
        if (workspace != nullptr) {                                // the caller's own lists (graph-safe, any stream)
            if (workspace_bytes < puct_workspace_bytes(R) || !aligned(workspace, 4)) return LZ_ERR_ARG;
            sc.lists = static_cast<int*>(workspace);
            sc.cap = R;
            sc.counts = reinterpret_cast<unsigned*>(sc.lists + kPuctLists * sc.cap);
            have = true;
        } else {
            have = puct_scratch(device, st, R, &sc);
        }
        if (!(nobin && nobin[0] == '0') && have) {
            hipLaunchKernelGGL(puct_zero_counts_kernel, dim3(1), dim3(64), 0, st, sc.counts);
            hipLaunchKernelGGL(puct_bin_kernel, grid, block, 0, st, valid, R, (int)A, sc.lists, sc.counts, sc.cap);
            if (A <= 64) hipLaunchKernelGGL(root_puct_binned_kernel<1>, grid, block, 0, st, priors, leaf, valid, R, (int)A, (int)sims, c, visits, value_sum, root_values, sc.lists, sc.counts, sc.cap, tables);
            else if (A <= 128) hipLaunchKernelGGL(root_puct_binned_kernel<2>, grid, block, 0, st, priors, leaf, valid, R, (int)A, (int)sims, c, visits, value_sum, root_values, sc.lists, sc.counts, sc.cap, tables);
            else hipLaunchKernelGGL(root_puct_binned_kernel<4>, grid, block, 0, st, priors, leaf, valid, R, (int)A, (int)sims, c, visits, value_sum, root_values, sc.lists, sc.counts, sc.cap, tables);
            return launch_status();
        }
        if (A <= 64) hipLaunchKernelGGL(root_puct_fast_kernel<1>, grid, block, 0, st, priors, leaf, valid, R, (int)A, (int)sims, c, visits, value_sum, root_values, tables);
        else if (A <= 128) hipLaunchKernelGGL(root_puct_fast_kernel<2>, grid, block, 0, st, priors, leaf, valid, R, (int)A, (int)sims, c, visits, value_sum, root_values, tables);
        else hipLaunchKernelGGL(root_puct_fast_kernel<4>, grid, block, 0, st, priors, leaf, valid, R, (int)A, (int)sims, c, visits, value_sum, root_values, tables);
        return launch_status();
    }
    if (A <= 64) hipLaunchKernelGGL(root_puct_kernel<1>, grid, block, 0, st, priors, leaf, valid, R, (int)A, sims, c, visits, value_sum, root_values);
    else if (A <= 128) hipLaunchKernelGGL(root_puct_kernel<2>, grid, block, 0, st, priors, leaf, valid, R, (int)A, sims, c, visits, value_sum, root_values);
    else hipLaunchKernelGGL(root_puct_kernel<4>, grid, block, 0, st, priors, leaf, valid, R, (int)A, sims, c, visits, value_sum, root_values);
    return launch_status();
}

int lz_root_puct_workspace_bytes(int64_t num_roots, int64_t* bytes) {
    if (num_roots < 0 || !bytes) return LZ_ERR_ARG;
    *bytes = puct_workspace_bytes(num_roots);
    return LZ_OK;
}

int lz_root_puct_allocate_visits(const float* priors, const float* leaf, const uint8_t* valid, int64_t R,
                                 int64_t A, int64_t sims, float c, float* visits, float* value_sum,
                                 float* root_values, void* stream) {
    return root_puct_impl(priors, leaf, valid, R, A, sims, c, visits, value_sum, root_values, nullptr, 0, stream);
}

int lz_root_puct_allocate_visits_ws(const float* priors, const float* leaf, const uint8_t* valid, int64_t R,
                                    int64_t A, int64_t sims, float c, float* visits, float* value_sum,
                                    float* root_values, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!workspace) return LZ_ERR_ARG;
    return root_puct_impl(priors, leaf, valid, R, A, sims, c, visits, value_sum, root_values, workspace, workspace_bytes,
                          stream);
}

int lz_root_finalize_from_visits(const int64_t* lidx, const int32_t* codes, const uint8_t* valid,
                                 const float* visits, const float* value_sum, const int64_t* roots, int64_t R,
                                 int64_t M, int64_t B, int64_t T, const float* temps, const float* uniforms,
                                 float* policy, int64_t* cidx, int32_t* ccodes, uint8_t* cvalid,
                                 float* root_value, void* stream) {
    if (B < 0 || T <= 0 || R < 0 || M < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!policy || !cidx || !ccodes || !cvalid) return LZ_ERR_ARG;
    if (R > 0 && M > 0 && (!lidx || !codes || !valid || !visits || !value_sum || !roots || !temps || !root_value))
        return LZ_ERR_ARG;
    if (M > 256) return LZ_ERR_UNSUPPORTED;
    if (!aligned(ccodes, 16) || (codes && !aligned(codes, 16))) return LZ_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    if (B > 0) {
        const int64_t work = B * T;
        const unsigned g = (unsigned)((work + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(finalize_fill_kernel, dim3(g > 2048u ? 2048u : (g ? g : 1u)), dim3(kBlock), 0, st, policy,
                           work, cidx, ccodes, cvalid, B);
    }
    if (R > 0 && M > 0) {
        const dim3 grid(grid_waves(R)), block(kBlock);
        const int4* c4 = reinterpret_cast<const int4*>(codes);
        int4* o4 = reinterpret_cast<int4*>(ccodes);
        if (M <= 64) hipLaunchKernelGGL(root_finalize_kernel<1>, grid, block, 0, st, lidx, c4, valid, visits, value_sum, roots, R, (int)M, B, (int)T, temps, uniforms, policy, cidx, o4, cvalid, root_value);
        else if (M <= 128) hipLaunchKernelGGL(root_finalize_kernel<2>, grid, block, 0, st, lidx, c4, valid, visits, value_sum, roots, R, (int)M, B, (int)T, temps, uniforms, policy, cidx, o4, cvalid, root_value);
        else hipLaunchKernelGGL(root_finalize_kernel<4>, grid, block, 0, st, lidx, c4, valid, visits, value_sum, roots, R, (int)M, B, (int)T, temps, uniforms, policy, cidx, o4, cvalid, root_value);
    }
    return launch_status();
}

int lz_root_force_uniform_picks(const int64_t* lidx, const int32_t* codes, const uint8_t* valid, const int64_t* roots,
                                int64_t R, int64_t M, const uint8_t* force_mask, const float* uniforms, int64_t* cidx,
                                int32_t* ccodes, uint8_t* cvalid, void* stream) {
    if (R < 0 || M < 0) return LZ_ERR_ARG;
    if (R == 0 || M == 0) return LZ_OK;
    if (!lidx || !codes || !valid || !force_mask || !uniforms || !cidx || !ccodes || !cvalid) return LZ_ERR_ARG;
    if (!aligned(codes, 16) || !aligned(ccodes, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(root_force_uniform_kernel, dim3(grid_waves(R)), dim3(kBlock), 0, as_stream(stream), lidx,
                       reinterpret_cast<const int4*>(codes), valid, roots, R, (int)M, force_mask, uniforms, cidx,
                       reinterpret_cast<int4*>(ccodes), cvalid);
    return launch_status();
}

int lz_self_play_step_inplace(const LzStateSoA* s, int64_t B, int64_t* plies, uint8_t* done,
                              const int64_t* active, int64_t n_active, const int32_t* codes,
                              const uint8_t* terminal, const uint8_t* cvalid, int64_t max_plies, float k,
                              int32_t* fin_kind, float* result, float* soft, void* stream) {
    if (B < 0 || n_active < 0 || max_plies <= 0) return LZ_ERR_ARG;
    if (n_active == 0) return LZ_OK;
    if (!soa_ok(s) || !plies || !done) return LZ_ERR_ARG;
    if (!active || !codes || !terminal || !cvalid || !fin_kind || !result || !soft) return LZ_ERR_ARG;
    if (!soa_aligned(s) || !aligned(codes, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(self_play_step_kernel, dim3(grid_threads(n_active)), dim3(kBlock), 0, as_stream(stream), *s, B,
                       plies, done, active, n_active, reinterpret_cast<const int4*>(codes), terminal, cvalid,
                       max_plies, k, fin_kind, result, soft);
    return launch_status();
}

int lz_finalize_trajectory_inplace(float* value_t, float* soft_t, const int8_t* signs, const int64_t* step_index,
                                   const int64_t* step_counts, int64_t G, int64_t Tmax, const int64_t* slots,
                                   const float* result, const float* softv, int64_t F, uint8_t* keep,
                                   int64_t* final_counts, int64_t* counts_out, void* stream) {
    if (F < 0 || G < 0 || Tmax < 0) return LZ_ERR_ARG;
    if (F == 0) return LZ_OK;
    if (!value_t || !soft_t || !signs || !step_index || !step_counts || !slots || !result || !softv || !keep ||
        !final_counts || !counts_out)
        return LZ_ERR_ARG;
    hipLaunchKernelGGL(finalize_trajectory_kernel, dim3(grid_waves(F)), dim3(kBlock), 0, as_stream(stream), value_t,
                       soft_t, signs, step_index, step_counts, G, Tmax, slots, result, softv, F, keep, final_counts,
                       reinterpret_cast<unsigned long long*>(counts_out));
    return launch_status();
}

int lz_wave_record(const uint8_t* done, int64_t G, int64_t* cursor, int64_t capacity, int64_t Tmax, int64_t* step_index,
                   int64_t* step_counts, int64_t* rows, int32_t* overflow, const float* model_input,
                   const uint8_t* legal_mask, const float* policy, const int64_t* current_player, int64_t T,
                   float* a_state, uint8_t* a_legal, float* a_policy, float* a_value, float* a_soft, int8_t* a_sign,
                   void* stream) {
    if (G < 0 || capacity < 0 || Tmax <= 0 || T <= 0) return LZ_ERR_ARG;
    if (G == 0) return LZ_OK;
    if (!done || !step_counts || !rows || !overflow || !model_input || !legal_mask || !policy ||
        !current_player || !a_state || !a_legal || !a_policy || !a_value || !a_soft || !a_sign)
        return LZ_ERR_ARG;
    if ((cursor == nullptr) != (step_index == nullptr)) return LZ_ERR_ARG;
    hipStream_t st = as_stream(stream);
    if (!cursor) {                                       // slot-major live arena: row = slot * Tmax + step
        if (capacity < G * Tmax) return LZ_ERR_ARG;
        hipLaunchKernelGGL(wave_rows_slot_kernel, dim3(grid_threads(G)), dim3(kBlock), 0, st, done, G, Tmax, step_counts,
                           rows, overflow);
    } else {
        hipLaunchKernelGGL(wave_rows_kernel, dim3(1), dim3(kRowsBlock), 0, st, done, G, cursor, capacity, Tmax, step_index,
                           step_counts, rows, overflow);
    }
    hipLaunchKernelGGL(wave_record_kernel, dim3(grid_waves(G)), dim3(kBlock), 0, st, rows, G, model_input, legal_mask,
                       policy, current_player, (int)T, a_state, a_legal, a_policy, a_value, a_soft, a_sign);
    return launch_status();
}

int lz_wave_step_finish(const LzStateSoA* s, int64_t G, int64_t* plies, uint8_t* done, const int32_t* codes,
                        const uint8_t* terminal, const uint8_t* cvalid, int64_t max_plies, float k, float* value_t,
                        float* soft_t, const int8_t* signs, const int64_t* step_index, int64_t* step_counts,
                        int64_t Tmax, int64_t* outcome, int64_t* delta_hist, int64_t* lengths,
                        const int64_t* slot_game, int64_t* finished, uint8_t* reseated, int reseat, void* stream) {
    if (G < 0 || max_plies <= 0 || Tmax <= 0) return LZ_ERR_ARG;
    if (G == 0) return LZ_OK;
    if (!soa_ok(s) || !plies || !done || !codes || !terminal || !cvalid || !value_t || !soft_t || !signs ||
        !step_counts || !outcome)
        return LZ_ERR_ARG;                                 // step_index == NULL: slot-major arena (row = g * Tmax + j)
    if (!soa_aligned(s) || !aligned(codes, 16)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(wave_step_finish_kernel, dim3(grid_waves(G)), dim3(kBlock), 0, as_stream(stream), *s, G, plies,
                       done, reinterpret_cast<const int4*>(codes), terminal, cvalid, max_plies, k, value_t, soft_t,
                       signs, step_index, step_counts, Tmax, reinterpret_cast<unsigned long long*>(outcome),
                       reinterpret_cast<unsigned long long*>(delta_hist), lengths, slot_game,
                       reinterpret_cast<unsigned long long*>(finished), reseated, reseat);
    return launch_status();
}

int lz_wave_reseat(const LzStateSoA* s, int64_t G, uint8_t* done, int64_t* plies, int64_t* step_counts, int64_t* budget,
                   int64_t* next_game, int64_t* slot_game, uint8_t* reseated, int logged_only, void* stream) {
    if (G < 0) return LZ_ERR_ARG;
    if (G == 0) return LZ_OK;
    if (!soa_ok(s) || !done || !plies || !step_counts || !budget || !next_game || !slot_game) return LZ_ERR_ARG;
    if (!soa_aligned(s)) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(wave_reseat_kernel, dim3(1), dim3(kRowsBlock), 0, as_stream(stream), *s, G, done, plies,
                       step_counts, budget, next_game, slot_game, reseated, logged_only);
    return launch_status();
}

int lz_wave_log_finished(const uint8_t* done, int64_t* step_counts, int64_t G, int64_t Tmax, int64_t T,
                         const float* a_state, const uint8_t* a_legal, const float* a_policy, const float* a_value,
                         const float* a_soft, float* l_state, uint8_t* l_legal, float* l_policy, float* l_value,
                         float* l_soft, int64_t log_capacity, int64_t* log_state, int64_t* log_base, void* stream) {
    if (G < 0 || Tmax <= 0 || T <= 0 || log_capacity < 0) return LZ_ERR_ARG;
    if (G == 0) return LZ_OK;
    if (!done || !step_counts || !a_state || !a_legal || !a_policy || !a_value || !a_soft || !l_state || !l_legal ||
        !l_policy || !l_value || !l_soft || !log_state || !log_base)
        return LZ_ERR_ARG;
    if (!aligned(a_state, 16) || !aligned(l_state, 16) || !aligned(a_policy, 16) || !aligned(l_policy, 16) ||
        !aligned(a_legal, 4) || !aligned(l_legal, 4))
        return LZ_ERR_ALIGN;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(wave_log_plan_kernel, dim3(1), dim3(kRowsBlock), 0, st, done, step_counts, G, log_capacity,
                       log_state, log_base);
    hipLaunchKernelGGL(wave_log_copy_kernel, dim3(grid_waves(G)), dim3(kBlock), 0, st, log_base, step_counts, G, Tmax,
                       (int)T, a_state, a_legal, a_policy, a_value, a_soft, l_state, l_legal, l_policy, l_value, l_soft);
    return launch_status();
}

}  // extern "C"
