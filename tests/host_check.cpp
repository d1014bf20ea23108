// tests/host_check.cpp -- TEST-ONLY host build of the bitboard rule header (liuzhou_amd/csrc/lz_rules.h)
// so its logic can be checked against the golden vectors on a machine without a GPU.  The product
// never loads this library; the GPU kernels include the same header.
#include <cstdint>
#include <cstring>
#include "../liuzhou_amd/csrc/lz_soa.h"
#include "../liuzhou_amd/csrc/lz_rng.h"

using namespace lz;

static RawState load_raw(const LzStateSoA* s, int64_t i) {
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
static State to_state(const RawState& r) {
    State s;
    s.black = r.black; s.white = r.white; s.mb = r.mb; s.mw = r.mw;
    s.phase = (int)r.phase; s.player = (int)r.player; s.pm_req = (int)r.pm_req; s.pm_rem = (int)r.pm_rem;
    s.pc_req = (int)r.pc_req; s.pc_rem = (int)r.pc_rem; s.forced = (int)r.forced;
    s.move_count = (int)r.move_count; s.msc = (int)r.msc;
    return s;
}
static void store(const LzStateSoA* o, int64_t i, const State& s) {
    for (int c = 0; c < 36; ++c) {
        o->board[i * 36 + c] = (s.black >> c) & 1 ? 1 : ((s.white >> c) & 1 ? -1 : 0);
        o->marks_black[i * 36 + c] = (s.mb >> c) & 1;
        o->marks_white[i * 36 + c] = (s.mw >> c) & 1;
    }
    o->phase[i] = s.phase; o->current_player[i] = s.player;
    o->pending_marks_required[i] = s.pm_req; o->pending_marks_remaining[i] = s.pm_rem;
    o->pending_captures_required[i] = s.pc_req; o->pending_captures_remaining[i] = s.pc_rem;
    o->forced_removals_done[i] = s.forced; o->move_count[i] = s.move_count; o->moves_since_capture[i] = s.msc;
}

extern "C" {
void hc_encode_actions(const LzStateSoA* s, int64_t B, int64_t ad, uint8_t* mask, int32_t* meta, int fallback) {
    const int T = 216 + (int)ad;
    for (int64_t i = 0; i < B; ++i) {
        RawState r = load_raw(s, i);
        Legal L = legal_actions(r.black, r.white, pick_cells(r, r.player), pick_cells(r, -r.player), r.empty,
                                r.mb, r.mw, (int)r.phase, (int)r.player, (int)r.pm_rem, (int)r.pc_rem,
                                (int)r.forced, fallback);
        for (int a = 0; a < T; ++a) {
            bool lg = legal_bit(L, a);
            mask[i * T + a] = lg;
            int k = -1, p = -1, q = -1, e = -1;
            if (lg) {
                index_to_code((int)r.phase, a, k, p, q, e);
                if (a >= 180 && a < 216) k = L.sel_kind;
            }
            int32_t* m = meta + (i * T + a) * 4;
            m[0] = k; m[1] = p; m[2] = q; m[3] = e;
        }
    }
}
void hc_apply_moves(const LzStateSoA* s, int64_t B, const int32_t* codes, const int64_t* parents, int64_t N,
                    const LzStateSoA* out) {
    for (int64_t i = 0; i < N; ++i) {
        if (parents[i] < 0 || parents[i] >= B) continue;
        State st = to_state(load_raw(s, parents[i]));
        apply(st, codes[i * 4], codes[i * 4 + 1], codes[i * 4 + 2]);
        // round-trip through the packed 32-byte engine form as well
        st = unpack(pack(st));
        store(out, i, st);
    }
}
void hc_status(const LzStateSoA* s, int64_t B, int32_t* status, int32_t* nlegal_py) {
    for (int64_t i = 0; i < B; ++i) {
        State st = to_state(load_raw(s, i));
        status[i] = game_status(st);
        nlegal_py[i] = legal_count(legal_actions(st, 0));
    }
}
// per-game counter RNG header (lz_rng.h), host build
void hc_philox(const uint32_t* ctr, uint32_t k0, uint32_t k1, int64_t n, uint32_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        const lzrng::U4 r = lzrng::philox4x32_10({ctr[i * 4], ctr[i * 4 + 1], ctr[i * 4 + 2], ctr[i * 4 + 3]}, k0, k1);
        out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
    }
}
void hc_rng_gamma(uint64_t seed, const int64_t* game, const int64_t* ply, int64_t B, float alpha, int64_t count, float* out) {
    for (int64_t g = 0; g < B; ++g)
        for (int64_t k = 0; k < count; ++k) out[g * count + k] = lzrng::gamma_draw(seed, game[g], ply[g], (uint32_t)k, alpha);
}
void hc_rng_uniform(uint64_t seed, const int64_t* game, const int64_t* ply, int64_t B, int purpose, float* out) {
    for (int64_t g = 0; g < B; ++g) out[g] = lzrng::uniform_draw(seed, game[g], ply[g], (uint32_t)purpose);
}
}
