// lz_host.cpp -- host (CPU-tensor) build of the v0_core OPERATOR subset of include/liuzhou_hip.h:
// libliuzhou_host.so exports the same entry points with the same signatures (`stream` is ignored), so that the
// Python `v0_core` module dispatches on the tensors' device exactly like the reference extension does
// (`board.device().is_cuda()`, v0/src/game/fast_legal_mask.cpp:453; CPU bodies v0/src/game/fast_legal_mask.cpp:253-418,
// fast_apply_moves.cpp:595-997, v0/src/net/encoding.cpp:26-79, project_policy_logits_fast.cpp:16-164,
// v0/src/bindings/module.cpp:180-871).  This is device dispatch, not a fallback: HIP tensors never come here, and the
// search engines / network kernel have no host build.  The rules are the same bitboard header the kernels include
// (lz_rules.h); everything is a plain loop over rows.  CPU error convention of the reference: an illegal action is an
// ERROR (LZ_ERR_ILLEGAL -> RuntimeError), where the GPU path is a silent no-op.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lz_soa.h"

using namespace lz;

namespace {

RawState load_raw(const LzStateSoA* s, int64_t i) {
    RawState r;
    r.black = cells_equal(s->board + i * 36, 1);
    r.white = cells_equal(s->board + i * 36, -1);
    r.empty = cells_equal(s->board + i * 36, 0);
    r.mb = cells_nonzero(s->marks_black + i * 36);
    r.mw = cells_nonzero(s->marks_white + i * 36);
    r.phase = s->phase[i]; r.player = s->current_player[i];
    r.pm_req = s->pending_marks_required[i]; r.pm_rem = s->pending_marks_remaining[i];
    r.pc_req = s->pending_captures_required[i]; r.pc_rem = s->pending_captures_remaining[i];
    r.forced = s->forced_removals_done[i]; r.move_count = s->move_count[i]; r.msc = s->moves_since_capture[i];
    return r;
}
State to_state(const RawState& r) {
    State s;
    s.black = r.black; s.white = r.white; s.mb = r.mb; s.mw = r.mw;
    s.phase = (int)r.phase; s.player = (int)r.player; s.pm_req = (int)r.pm_req; s.pm_rem = (int)r.pm_rem;
    s.pc_req = (int)r.pc_req; s.pc_rem = (int)r.pc_rem; s.forced = (int)r.forced;
    s.move_count = (int)r.move_count; s.msc = (int)r.msc;
    return s;
}
void store(const LzStateSoA* o, int64_t i, const State& s) {
    for (int c = 0; c < 36; ++c) {
        o->board[i * 36 + c] = (int8_t)(((s.black >> c) & 1) ? 1 : (((s.white >> c) & 1) ? -1 : 0));
        o->marks_black[i * 36 + c] = (uint8_t)((s.mb >> c) & 1);
        o->marks_white[i * 36 + c] = (uint8_t)((s.mw >> c) & 1);
    }
    o->phase[i] = s.phase; o->current_player[i] = s.player;
    o->pending_marks_required[i] = s.pm_req; o->pending_marks_remaining[i] = s.pm_rem;
    o->pending_captures_required[i] = s.pc_req; o->pending_captures_remaining[i] = s.pc_rem;
    o->forced_removals_done[i] = s.forced; o->move_count[i] = s.move_count; o->moves_since_capture[i] = s.msc;
}
bool soa_ok(const LzStateSoA* s) {
    return s && s->board && s->marks_black && s->marks_white && s->phase && s->current_player &&
           s->pending_marks_required && s->pending_marks_remaining && s->pending_captures_required &&
           s->pending_captures_remaining && s->forced_removals_done && s->move_count && s->moves_since_capture;
}
Legal legal_of(const RawState& r, int fallback) {
    return legal_actions(r.black, r.white, pick_cells(r, r.player), pick_cells(r, -r.player), r.empty, r.mb, r.mw,
                         (int)r.phase, (int)r.player, (int)r.pm_rem, (int)r.pc_rem, (int)r.forced, fallback);
}
float soft_value(uint64_t black, uint64_t white, float k) {
    return std::tanh((float)(popc(black) - popc(white)) / 18.0f * k);
}

}  // namespace

extern "C" {

const char* lz_version(void) { return "liuzhou-host 0.1 (cpu tensors)"; }

const char* lz_status_string(int status) {
    switch (status) {
        case LZ_OK: return "ok";
        case LZ_ERR_ARG: return "invalid argument";
        case LZ_ERR_UNSUPPORTED: return "unsupported dimensions";
        case LZ_ERR_LAUNCH: return "kernel launch failed";
        case LZ_ERR_ALIGN: return "misaligned pointer";
        case LZ_ERR_ILLEGAL: return "illegal action for the state (CPU tensors: the reference's CPU path raises too)";
        default: return "unknown status";
    }
}

// fast_legal_mask.cpp:253-418
int lz_encode_actions_fast(const LzStateSoA* s, int64_t B, int64_t pd, int64_t md, int64_t sd, int64_t ad,
                           uint8_t* mask, int32_t* meta, void*) {
    if (B < 0 || ad < 0) return LZ_ERR_ARG;
    if (pd != 36 || md != 144 || sd != 36 || ad > 40) return LZ_ERR_UNSUPPORTED;
    if (B == 0) return LZ_OK;
    if (!soa_ok(s) || !mask || !meta) return LZ_ERR_ARG;
    const int T = 216 + (int)ad;
    for (int64_t i = 0; i < B; ++i) {
        const RawState r = load_raw(s, i);
        const Legal L = legal_of(r, /*fallback_forced=*/1);
        for (int a = 0; a < T; ++a) {
            const bool lg = a < 217 && legal_bit(L, a);
            mask[i * T + a] = lg ? 1 : 0;
            int k = -1, p = -1, q = -1, e = -1;
            if (lg) {
                index_to_code((int)r.phase, a, k, p, q, e);
                if (a >= 180 && a < 216) k = L.sel_kind;
            }
            int32_t* m = meta + (i * T + a) * 4;
            m[0] = k; m[1] = p; m[2] = q; m[3] = e;
        }
    }
    return LZ_OK;
}

// fast_apply_moves.cpp:595-938 (the CPU path checks every action: TORCH_CHECK -> here LZ_ERR_ILLEGAL)
int lz_batch_apply_moves(const LzStateSoA* in, int64_t B, const int32_t* codes, const int64_t* parents, int64_t N,
                         const LzStateSoA* out, void*) {
    if (B < 0 || N < 0) return LZ_ERR_ARG;
    if (N == 0) return LZ_OK;
    if (!soa_ok(in) || !soa_ok(out) || !codes || !parents) return LZ_ERR_ARG;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t p = parents[i];
        if (p < 0 || p >= B) return LZ_ERR_ARG;
        State st = to_state(load_raw(in, p));
        if (!apply(st, codes[i * 4], codes[i * 4 + 1], codes[i * 4 + 2])) return LZ_ERR_ILLEGAL;
        store(out, i, st);
    }
    return LZ_OK;
}

int lz_batch_apply_moves_inplace(const LzStateSoA* s, int64_t B, const int32_t* codes, const int64_t* slots, int64_t N,
                                 void*) {
    if (B < 0 || N < 0) return LZ_ERR_ARG;
    if (N == 0) return LZ_OK;
    if (!soa_ok(s) || !codes || !slots) return LZ_ERR_ARG;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t p = slots[i];
        if (p < 0 || p >= B) return LZ_ERR_ARG;
        State st = to_state(load_raw(s, p));
        if (!apply(st, codes[i * 4], codes[i * 4 + 1], codes[i * 4 + 2])) return LZ_ERR_ILLEGAL;
        store(s, p, st);
    }
    return LZ_OK;
}

// encoding.cpp:26-79 (the player is compared as int8, like the reference's cast)
int lz_states_to_model_input(const int8_t* board, const uint8_t* mb, const uint8_t* mw, const int64_t* phase,
                             const int64_t* player, int64_t B, float* out, void*) {
    if (B < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!board || !mb || !mw || !phase || !player || !out) return LZ_ERR_ARG;
    for (int64_t i = 0; i < B; ++i) {
        const int cur8 = (int)(int8_t)player[i], neg8 = (int)(int8_t)(-(int8_t)player[i]);
        const bool black = player[i] == 1;
        float* o = out + i * 396;
        for (int c = 0; c < 36; ++c) {
            const int v = board[i * 36 + c];
            const bool m1 = mb[i * 36 + c] != 0, m2 = mw[i * 36 + c] != 0;
            o[c] = v == cur8 ? 1.f : 0.f;
            o[36 + c] = v == neg8 ? 1.f : 0.f;
            o[72 + c] = (black ? m1 : m2) ? 1.f : 0.f;
            o[108 + c] = (black ? m2 : m1) ? 1.f : 0.f;
            for (int ph = 1; ph <= 7; ++ph) o[(3 + ph) * 36 + c] = phase[i] == ph ? 1.f : 0.f;
        }
    }
    return LZ_OK;
}

// project_policy_logits_fast.cpp:16-164
int lz_project_policy_logits_fast(const float* lp1, const float* lp2, const float* lpmc, const uint8_t* mask, int64_t B,
                                  int64_t pd, int64_t md, int64_t sd, int64_t ad, float* probs, float* masked_logits,
                                  void*) {
    if (B < 0 || ad < 0) return LZ_ERR_ARG;
    if (pd != 36 || md != 144 || sd != 36 || ad > 40) return LZ_ERR_UNSUPPORTED;
    if (B == 0) return LZ_OK;
    if (!lp1 || !lp2 || !lpmc || !mask || !probs || !masked_logits) return LZ_ERR_ARG;
    const int T = 216 + (int)ad;
    std::vector<float> v((size_t)T);
    for (int64_t row = 0; row < B; ++row) {
        const float* h1 = lp1 + row * 36; const float* h2 = lp2 + row * 36; const float* hm = lpmc + row * 36;
        float mx = -INFINITY;
        bool any_legal = false, any_finite = false;
        for (int a = 0; a < T; ++a) {
            float x;
            if (a < 36) x = h1[a];
            else if (a < 180) {
                const int from = (a - 36) >> 2, d = (a - 36) & 3;
                const int r = from / 6, c = from - 6 * r;
                const bool on_board = !((d == 0 && r == 0) || (d == 1 && r == 5) || (d == 2 && c == 0) || (d == 3 && c == 5));
                x = on_board ? h2[from] + h1[move_dest(from, d)] : -INFINITY;
            } else if (a < 216) x = hm[a - 180];
            else x = 0.f;
            const bool legal = mask[row * T + a] != 0;
            v[a] = legal ? x : -INFINITY;
            any_legal = any_legal || legal;
            any_finite = any_finite || std::isfinite(v[a]);
            if (v[a] > mx) mx = v[a];
        }
        const bool do_softmax = any_legal && any_finite;
        float sum = 0.f;
        if (do_softmax)
            for (int a = 0; a < T; ++a) sum += v[a] == -INFINITY ? 0.f : std::exp(v[a] - mx);
        for (int a = 0; a < T; ++a) {
            float pr = 0.f, ml = v[a];
            if (do_softmax) pr = (v[a] == -INFINITY ? 0.f : std::exp(v[a] - mx)) / sum;
            else if (any_legal && mask[row * T + a] != 0) ml = 0.f;      // project_policy_logits_fast.cpp:153-160
            probs[row * T + a] = pr;
            masked_logits[row * T + a] = ml;
        }
    }
    return LZ_OK;
}

// module.cpp:247-363, same three-step protocol as the device library
int lz_root_pack_rows(const uint8_t* mask, const float* probs, const int32_t* meta, int64_t B, int64_t T, int64_t cap,
                      int32_t* counts, int32_t* legal_index, float* priors, int32_t* codes, void*) {
    if (B < 0 || T <= 0 || cap <= 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!mask || !probs || !meta || !counts || !legal_index || !priors || !codes) return LZ_ERR_ARG;
    for (int64_t row = 0; row < B; ++row) {
        float part = 0.f;
        for (int64_t a = 0; a < T; ++a) if (mask[row * T + a]) part += probs[row * T + a];
        const float denom = part > 1e-8f ? part : 1e-8f;
        for (int64_t j = 0; j < cap; ++j) {
            legal_index[row * cap + j] = -1; priors[row * cap + j] = 0.f;
            std::memset(codes + (row * cap + j) * 4, 0, 16);
        }
        int n = 0;
        for (int64_t a = 0; a < T; ++a) {
            if (!mask[row * T + a]) continue;
            if (n < cap) {
                legal_index[row * cap + n] = (int32_t)a;
                priors[row * cap + n] = probs[row * T + a] / denom;
                std::memcpy(codes + (row * cap + n) * 4, meta + (row * T + a) * 4, 16);
            }
            ++n;
        }
        counts[row] = n;
    }
    return LZ_OK;
}

int lz_root_pack_plan(const int32_t* counts, int64_t B, int32_t* rank, int64_t* child_off, int64_t* sizes, void*) {
    if (B < 0 || !sizes) return LZ_ERR_ARG;
    if (B > 0 && (!counts || !rank || !child_off)) return LZ_ERR_ARG;
    int64_t r = 0, k = 0, mx = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int c = counts[b];
        rank[b] = c > 0 ? (int32_t)r : -1;
        child_off[b] = k;
        if (c > 0) { ++r; k += c; }
        if (c > mx) mx = c;
    }
    sizes[0] = r; sizes[1] = mx; sizes[2] = k;
    return LZ_OK;
}

int lz_root_pack_fill(const int32_t* counts, const int32_t* legal_index, const float* priors, const int32_t* codes,
                      const int32_t* rank, const int64_t* child_off, int64_t B, int64_t cap, int64_t R, int64_t M,
                      int64_t N, uint8_t* terminal_mask, int64_t* valid_root_indices, int64_t* counts_out,
                      uint8_t* valid_mask, int64_t* legal_index_mat, float* priors_mat, int32_t* action_code_mat,
                      int64_t* pack_flat_idx, int32_t* action_codes_all, int64_t* parent_indices_all, void*) {
    if (B < 0 || cap < 1 || R < 0 || M < 0 || N < 0 || R > B || M > cap) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!counts || !legal_index || !priors || !codes || !rank || !child_off || !terminal_mask) return LZ_ERR_ARG;
    for (int64_t b = 0; b < B; ++b) {
        const int c = counts[b];
        const int r = rank[b];
        terminal_mask[b] = c == 0 ? 1 : 0;
        if (r < 0) continue;
        valid_root_indices[r] = b; counts_out[r] = c;
        const int64_t off = child_off[b];
        for (int64_t k = 0; k < M; ++k) {
            const bool ok = k < c && k < cap;
            const int64_t o = (int64_t)r * M + k;
            valid_mask[o] = ok ? 1 : 0;
            const int li = ok ? legal_index[b * cap + k] : 0;
            legal_index_mat[o] = li < 0 ? 0 : li;
            priors_mat[o] = ok ? priors[b * cap + k] : 0.f;
            if (ok) std::memcpy(action_code_mat + o * 4, codes + (b * cap + k) * 4, 16);
            else std::memset(action_code_mat + o * 4, 0, 16);
            if (ok) {
                pack_flat_idx[off + k] = o;
                std::memcpy(action_codes_all + (off + k) * 4, codes + (b * cap + k) * 4, 16);
                parent_indices_all[off + k] = b;
            }
        }
    }
    (void)N;
    return LZ_OK;
}

// module.cpp:180-245 == root_puct_fused.cu:44-116: fp32, same operation order, first maximum wins
int lz_root_puct_allocate_visits(const float* priors, const float* leaf, const uint8_t* valid, int64_t R, int64_t A,
                                 int64_t sims, float c, float* visits, float* value_sum, float* root_values, void*) {
    if (R < 0 || A < 0 || sims <= 0) return LZ_ERR_ARG;
    if (R == 0 || A == 0) return LZ_OK;
    if (!priors || !leaf || !valid || !visits || !value_sum || !root_values) return LZ_ERR_ARG;
    for (int64_t r = 0; r < R; ++r) {
        float* vis = visits + r * A; float* vs = value_sum + r * A;
        for (int64_t a = 0; a < A; ++a) { vis[a] = 0.f; vs[a] = 0.f; }
        float total = 0.f;
        for (int64_t sim = 0; sim < sims; ++sim) {
            const float sqrt_total = std::sqrt(total + 1.0f);
            float best = -INFINITY;
            int64_t pick = -1;
            for (int64_t a = 0; a < A; ++a) {
                if (!valid[r * A + a]) continue;
                const float q = vs[a] / (vis[a] > 1e-8f ? vis[a] : 1e-8f);
                const float u = c * priors[r * A + a] * sqrt_total / (1.0f + vis[a]);
                const float score = q + u;
                if (score == score && (pick < 0 || score > best)) { best = score; pick = a; }
            }
            if (pick >= 0) { vis[pick] += 1.0f; vs[pick] += leaf[r * A + pick]; }
            total += 1.0f;
        }
        float sv = 0.f, sw = 0.f;
        for (int64_t a = 0; a < A; ++a) { sv += vis[a]; sw += vs[a]; }
        root_values[r] = sw / (sv > 1.0f ? sv : 1.0f);
    }
    return LZ_OK;
}

// module.cpp:441-535 (+ the stable sampling of mcts_gpu.py:853-898 when `uniforms` is given)
int lz_root_finalize_from_visits(const int64_t* lidx, const int32_t* codes, const uint8_t* valid, const float* visits,
                                 const float* value_sum, const int64_t* roots, int64_t R, int64_t M, int64_t B, int64_t T,
                                 const float* temps, const float* uniforms, float* policy, int64_t* cidx, int32_t* ccodes,
                                 uint8_t* cvalid, float* root_value, void*) {
    if (B < 0 || T <= 0 || R < 0 || M < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!policy || !cidx || !ccodes || !cvalid) return LZ_ERR_ARG;
    for (int64_t j = 0; j < B * T; ++j) policy[j] = 0.f;
    for (int64_t b = 0; b < B; ++b) { cidx[b] = -1; cvalid[b] = 0; for (int k = 0; k < 4; ++k) ccodes[b * 4 + k] = -1; }
    if (R == 0 || M == 0) return LZ_OK;
    if (!lidx || !codes || !valid || !visits || !value_sum || !roots || !temps || !root_value) return LZ_ERR_ARG;
    std::vector<float> pol((size_t)M), ex((size_t)M);
    for (int64_t r = 0; r < R; ++r) {
        const int64_t b = roots[r];
        const float temp = temps[r] > 1e-6f ? temps[r] : 1e-6f;
        const float inv_t = 1.0f / temp;
        float psum = 0.f, sv = 0.f, sw = 0.f;
        bool any = false;
        for (int64_t a = 0; a < M; ++a) {
            const bool ok = valid[r * M + a] != 0;
            const float v = visits[r * M + a];
            pol[a] = ok ? std::pow(v > 1e-8f ? v : 1e-8f, inv_t) : 0.f;
            psum += pol[a]; sv += v; sw += value_sum[r * M + a];
            any = any || ok;
        }
        if (!any) continue;
        psum = psum > 1e-8f ? psum : 1e-8f;
        int64_t pick = 0;
        float best = -INFINITY;
        for (int64_t a = 0; a < M; ++a) {
            pol[a] = pol[a] / psum;
            if (pol[a] == pol[a] && pol[a] > best) { best = pol[a]; pick = a; }
        }
        if (uniforms != nullptr && M > 1) {
            float mx = -INFINITY;
            for (int64_t a = 0; a < M; ++a) {
                float v = visits[r * M + a];
                if (!(v == v) || std::isinf(v)) v = 0.f;
                ex[a] = valid[r * M + a] ? std::log(v > 1e-8f ? v : 1e-8f) * inv_t : -INFINITY;
                if (ex[a] > mx) mx = ex[a];
            }
            if (!std::isfinite(mx)) mx = 0.f;
            float esum = 0.f;
            int nvalid = 0;
            for (int64_t a = 0; a < M; ++a) {
                const bool ok = valid[r * M + a] != 0;
                float e = ok ? std::exp(ex[a] - mx) : 0.f;
                if (!(e == e) || std::isinf(e)) e = 0.f;
                ex[a] = e; esum += e; nvalid += ok ? 1 : 0;
            }
            if (!(esum > 0.f) || !std::isfinite(esum)) {
                for (int64_t a = 0; a < M; ++a) ex[a] = valid[r * M + a] ? 1.0f : 0.f;
                esum = (float)(nvalid > 0 ? nvalid : 1);
            }
            const float target = uniforms[r] * esum;
            float run = 0.f;
            int64_t chosen = -1, last = -1;
            for (int64_t a = 0; a < M; ++a) {
                run += ex[a];
                const bool live = valid[r * M + a] && ex[a] > 0.f;
                if (live) last = a;
                if (chosen < 0 && live && run > target) chosen = a;
            }
            if (chosen < 0) chosen = last;
            if (chosen >= 0) pick = chosen;
        }
        for (int64_t a = 0; a < M; ++a) {
            if (!valid[r * M + a]) continue;
            const int64_t col = lidx[r * M + a];
            if (col >= 0 && col < T) policy[b * T + col] += pol[a];
        }
        cidx[b] = lidx[r * M + pick];
        std::memcpy(ccodes + b * 4, codes + (r * M + pick) * 4, 16);
        cvalid[b] = 1;
        root_value[r] = sw / (sv > 1.0f ? sv : 1.0f);
    }
    return LZ_OK;
}

// module.cpp:632-871
int lz_self_play_step_inplace(const LzStateSoA* s, int64_t B, int64_t* plies, uint8_t* done, const int64_t* active,
                              int64_t n_active, const int32_t* codes, const uint8_t* terminal, const uint8_t* cvalid,
                              int64_t max_plies, float k, int32_t* fin_kind, float* result, float* soft, void*) {
    if (B < 0 || n_active < 0 || max_plies <= 0) return LZ_ERR_ARG;
    if (n_active == 0) return LZ_OK;
    if (!soa_ok(s) || !plies || !done || !active || !codes || !terminal || !cvalid || !fin_kind || !result || !soft)
        return LZ_ERR_ARG;
    for (int64_t i = 0; i < n_active; ++i) {
        const int64_t slot = active[i];
        if (slot < 0 || slot >= B) { fin_kind[i] = 0; result[i] = 0.f; soft[i] = 0.f; continue; }
        State st = to_state(load_raw(s, slot));
        const bool term = terminal[i] != 0;
        if (term || cvalid[i] == 0) {
            done[slot] = 1; fin_kind[i] = 1;
            result[i] = term ? -(float)s->current_player[slot] : 0.f;
            soft[i] = soft_value(st.black, st.white, k);
            continue;
        }
        if (!apply(st, codes[i * 4], codes[i * 4 + 1], codes[i * 4 + 2])) return LZ_ERR_ILLEGAL;
        store(s, slot, st);
        const int64_t np = plies[slot] + 1;
        plies[slot] = np;
        int winner = 0;
        const bool post = st.phase == kMovement || st.phase == kCaptureSelection || st.phase == kCounterRemoval;
        if (post && popc(st.black) < kLoseThreshold) winner = -1;
        if (post && popc(st.white) < kLoseThreshold) winner = 1;
        const bool draw = st.move_count >= kMaxMoveCount || st.msc >= kNoCaptureLimit;
        if (winner != 0 || draw || np >= max_plies) {
            done[slot] = 1; fin_kind[i] = 2; result[i] = (float)winner; soft[i] = soft_value(st.black, st.white, k);
        } else { fin_kind[i] = 0; result[i] = 0.f; soft[i] = 0.f; }
    }
    return LZ_OK;
}

// module.cpp:547-630
int lz_finalize_trajectory_inplace(float* value_t, float* soft_t, const int8_t* signs, const int64_t* step_index,
                                   const int64_t* step_counts, int64_t G, int64_t Tmax, const int64_t* slots,
                                   const float* result, const float* softv, int64_t F, uint8_t* keep,
                                   int64_t* final_counts, int64_t* counts_out, void*) {
    if (F < 0 || G < 0 || Tmax < 0) return LZ_ERR_ARG;
    if (F == 0) return LZ_OK;
    if (!value_t || !soft_t || !signs || !step_index || !step_counts || !slots || !result || !softv || !keep ||
        !final_counts || !counts_out)
        return LZ_ERR_ARG;
    for (int64_t f = 0; f < F; ++f) {
        const int64_t g = slots[f];
        int64_t n = (g >= 0 && g < G) ? step_counts[g] : 0;
        if (n > Tmax) n = Tmax;
        keep[f] = n > 0 ? 1 : 0;
        final_counts[f] = n;
        if (n > 0) counts_out[result[f] > 0.f ? 0 : (result[f] < 0.f ? 1 : 2)] += 1;
        for (int64_t j = 0; j < n; ++j) {
            const int64_t idx = step_index[g * Tmax + j];
            const float sg = (float)signs[idx];
            value_t[idx] = sg * result[f];
            soft_t[idx] = sg * softv[f];
        }
    }
    return LZ_OK;
}

}  // extern "C"
