// lz_scalar.cpp -- the scalar (one state per call) rule surface of include/liuzhou_scalar.h, part of libliuzhou_host.so.
//
// What the reference does with per-cell loops over a 6x6 array and std::vector pools (v0/src/rules/rule_engine.cpp,
// v0/src/moves/move_generator.cpp) is done here on the 36-bit bitboards of lz_rules.h -- the header the gfx950 kernels
// include -- so the host tools and the device engines cannot drift apart: a target pool is one `prefer_normal`, a shape
// test one `in_shape_set`, a transition one `apply_rule_t`.  This file adds what the scalar functions have on top of
// the batched operators: their own argument validation (every `throw` of the reference is a LzScalarReason here), the
// lists in the reference's order, and the two places where the scalar functions and the tensor operators of the
// reference differ on unreachable states (noted below).
#include <cstdint>
#include <cstring>

#include "../../include/liuzhou_hip.h"
#include "../../include/liuzhou_scalar.h"
#include "lz_rules.h"

using namespace lz;

namespace {

enum : int { kTypePlace = 1, kTypeMove = 2, kTypeMark = 3, kTypeCapture = 4, kTypeForced = 5, kTypeCounter = 6,
             kTypeNoMoves = 7, kTypeProcess = 8 };                    // v0::ActionType (move_generator.hpp:14-23)

State load(const LzScalarState& g) {
    State s;
    s.black = s.white = s.mb = s.mw = 0;
    for (int c = 0; c < kCells; ++c) {
        if (g.board[c] == 1) s.black |= 1ull << c;
        else if (g.board[c] == -1) s.white |= 1ull << c;
        if (g.marks_black[c]) s.mb |= 1ull << c;
        if (g.marks_white[c]) s.mw |= 1ull << c;
    }
    s.phase = g.phase; s.player = g.current_player < 0 ? -1 : 1;
    s.pm_req = g.pending_marks_required; s.pm_rem = g.pending_marks_remaining;
    s.pc_req = g.pending_captures_required; s.pc_rem = g.pending_captures_remaining;
    s.forced = g.forced_removals_done; s.move_count = g.move_count; s.msc = g.moves_since_capture;
    return s;
}

void store(const State& s, LzScalarState& g) {
    for (int c = 0; c < kCells; ++c) {
        g.board[c] = (int8_t)(((s.black >> c) & 1) ? 1 : (((s.white >> c) & 1) ? -1 : 0));
        g.marks_black[c] = (uint8_t)((s.mb >> c) & 1);
        g.marks_white[c] = (uint8_t)((s.mw >> c) & 1);
    }
    g.phase = s.phase; g.current_player = s.player;
    g.pending_marks_required = s.pm_req; g.pending_marks_remaining = s.pm_rem;
    g.pending_captures_required = s.pc_req; g.pending_captures_remaining = s.pc_rem;
    g.forced_removals_done = s.forced; g.move_count = s.move_count; g.moves_since_capture = s.msc;
}

inline bool on_board(int cell) { return cell >= 0 && cell < kCells; }
inline uint64_t bit_of(int cell) { return 1ull << cell; }
inline uint64_t& side(State& s, int player) { return player == 1 ? s.black : s.white; }
inline uint64_t side(const State& s, int player) { return player == 1 ? s.black : s.white; }
inline uint64_t marks_of(const State& s, int player) { return player == 1 ? s.mb : s.mw; }

int why(int32_t* reason, int code) {
    if (reason) *reason = code;
    return LZ_ERR_ILLEGAL;
}

// ---- the lists -------------------------------------------------------------------------------------------------
struct Out {
    int32_t* p; int32_t cap; int32_t n; bool overflow;
    void put(int32_t v) { if (n < cap) p[n++] = v; else overflow = true; }
    void cells(uint64_t set) { for (uint64_t m = set & kFull; m; m &= m - 1) put(ctz(m)); }
    void moves(int phase, int type, uint64_t set) {
        for (uint64_t m = set & kFull; m; m &= m - 1) { put(phase); put(type); put(ctz(m)); put(-1); }
    }
};

uint64_t mark_targets(const State& s) {            // rule_engine.cpp GenerateMarkTargets
    if (s.phase != kMarkSelection || s.pm_rem <= 0) return 0;
    const uint64_t opp = side(s, -s.player), om = marks_of(s, -s.player);
    return prefer_normal(opp & ~om, opp, opp & ~om);
}
uint64_t capture_targets(const State& s) {         // rule_engine.cpp GenerateCaptureTargets (candidacy ignores marks)
    if (s.phase != kCaptureSelection || s.pc_rem <= 0) return 0;
    const uint64_t opp = side(s, -s.player), om = marks_of(s, -s.player);
    return prefer_normal(opp, opp, opp & ~om);
}
uint64_t forced_targets(const State& s) {          // move_generator.cpp GenerateForcedRemovalMoves: no fall-back pool
    if (s.phase != kForcedRemoval || s.forced < 0 || s.forced > 1) return 0;
    const uint64_t tgt = s.forced == 0 ? s.black : s.white;
    return tgt & ~in_shape_set(tgt, tgt);
}
uint64_t removal_pool(const State& s, int phase) { // GenerateNoMovesOptions / GenerateCounterRemovalMoves: marks ignored
    if (s.phase != phase) return 0;
    const uint64_t opp = side(s, -s.player);
    return prefer_normal(opp, opp, opp);
}
void movement_moves(const State& s, Out& o, bool as_records) {
    if (s.phase != kMovement) return;
    const Legal L = legal_actions(s, 0);
    for (uint64_t m = (L.up | L.down | L.left | L.right) & kFull; m; m &= m - 1) {
        const int from = ctz(m);
        for (int d = 0; d < 4; ++d)
            if ((move_set(L, d) >> from) & 1) {
                if (as_records) { o.put(kMovement); o.put(kTypeMove); }
                o.put(from); o.put(move_dest(from, d));
            }
    }
}
bool any_movement(const State& s) {
    const Legal L = legal_actions(s, 0);
    return ((L.up | L.down | L.left | L.right) & kFull) != 0;
}

// ---- the transitions: validation in the reference's order, then the shared bitboard transition ------------------
int step(State& s, int what, int a, int b, int32_t* reason) {
    switch (what) {
    case LZ_STEP_PLACEMENT: {
        if (s.phase != kPlacement) return why(reason, LZ_WHY_PHASE);
        if (!on_board(a)) return why(reason, LZ_WHY_OUT_OF_BOARD);
        if ((s.black | s.white) & bit_of(a)) return why(reason, LZ_WHY_OCCUPIED);
        if (marks_of(s, -s.player) & bit_of(a)) return why(reason, LZ_WHY_MARKED_BY_OPPONENT);
        apply_rule_t<false>(s, kActPlace, a, -1);
        return LZ_OK;
    }
    case LZ_STEP_MARK: {
        if (s.phase != kMarkSelection) return why(reason, LZ_WHY_PHASE);
        if (s.pm_rem <= 0) return why(reason, LZ_WHY_NOTHING_PENDING);
        if (!on_board(a)) return why(reason, LZ_WHY_OUT_OF_BOARD);
        const uint64_t opp = side(s, -s.player), om = marks_of(s, -s.player);
        if (!(opp & bit_of(a))) return why(reason, LZ_WHY_NOT_OPPONENT_PIECE);
        if (om & bit_of(a)) return why(reason, LZ_WHY_ALREADY_MARKED);
        const uint64_t shaped = in_shape_set(opp, opp & ~om);
        if ((shaped & bit_of(a)) && (opp & ~om & ~shaped)) return why(reason, LZ_WHY_IN_SHAPE);
        apply_rule_t<false>(s, kActMark, a, -1);
        return LZ_OK;
    }
    case LZ_STEP_PROCESS_REMOVAL: {
        if (s.phase != kRemoval) return why(reason, LZ_WHY_PHASE);
        const uint64_t marked = (s.mb | s.mw) & kFull;
        if (marked && !(marked & (s.black | s.white))) {
            // marks on empty cells only (unreachable): the scalar function clears the marks and STAYS in the removal
            // phase (rule_engine.cpp: `if (removed > 0)`); the tensor operator moves on -- each keeps its own behaviour
            s.mb = s.mw = 0;
            return LZ_OK;
        }
        apply_rule_t<false>(s, kActProcess, -1, -1);
        return LZ_OK;
    }
    case LZ_STEP_MOVEMENT: {
        if (s.phase != kMovement) return why(reason, LZ_WHY_PHASE);
        if (!on_board(a) || !on_board(b)) return why(reason, LZ_WHY_OUT_OF_BOARD);
        if (!(side(s, s.player) & bit_of(a))) return why(reason, LZ_WHY_NOT_OWN_PIECE);
        if ((s.black | s.white) & bit_of(b)) return why(reason, LZ_WHY_OCCUPIED);
        const int ra = a / 6, ca = a - 6 * ra, rb = b / 6, cb = b - 6 * rb;
        int dir = -1;
        if (ca == cb && rb == ra - 1) dir = 0;
        else if (ca == cb && rb == ra + 1) dir = 1;
        else if (ra == rb && cb == ca - 1) dir = 2;
        else if (ra == rb && cb == ca + 1) dir = 3;
        if (dir < 0) return why(reason, LZ_WHY_NOT_ONE_STEP);
        apply_rule_t<false>(s, kActMove, a, dir);
        return LZ_OK;
    }
    case LZ_STEP_CAPTURE: {
        if (s.phase != kCaptureSelection) return why(reason, LZ_WHY_PHASE);
        if (s.pc_rem <= 0) return why(reason, LZ_WHY_NOTHING_PENDING);
        if (!on_board(a)) return why(reason, LZ_WHY_OUT_OF_BOARD);
        const uint64_t opp = side(s, -s.player), om = marks_of(s, -s.player);
        if (!(opp & bit_of(a))) return why(reason, LZ_WHY_NOT_OPPONENT_PIECE);
        const uint64_t shaped = in_shape_set(opp, opp & ~om);
        if ((shaped & bit_of(a)) && (opp & ~shaped)) return why(reason, LZ_WHY_IN_SHAPE);
        apply_rule_t<false>(s, kActCapture, a, -1);
        return LZ_OK;
    }
    case LZ_STEP_FORCED_REMOVAL: {
        if (s.phase != kForcedRemoval) return why(reason, LZ_WHY_PHASE);
        if (!on_board(a)) return why(reason, LZ_WHY_OUT_OF_BOARD);
        if (s.forced != 0 && s.forced != 1) return why(reason, LZ_WHY_FORCED_ORDER);
        const int remover = s.forced == 0 ? -1 : 1;
        if (s.player != remover) return why(reason, LZ_WHY_FORCED_ORDER);
        const uint64_t tgt = side(s, -remover);
        if (!(tgt & bit_of(a))) return why(reason, LZ_WHY_NOT_OPPONENT_PIECE);
        if (in_shape_set(tgt, tgt) & bit_of(a)) return why(reason, LZ_WHY_IN_SHAPE);
        apply_rule_t<false>(s, kActForced, a, -1);
        return LZ_OK;
    }
    case LZ_STEP_NO_MOVES:
    case LZ_STEP_COUNTER_REMOVAL: {
        const bool counter = what == LZ_STEP_COUNTER_REMOVAL;
        if (s.phase != (counter ? kCounterRemoval : kMovement)) return why(reason, LZ_WHY_PHASE);
        if (!on_board(a)) return why(reason, LZ_WHY_OUT_OF_BOARD);
        const uint64_t opp = side(s, -s.player);
        if (!(opp & bit_of(a))) return why(reason, LZ_WHY_NOT_OPPONENT_PIECE);
        const uint64_t shaped = in_shape_set(opp, opp);
        if ((shaped & bit_of(a)) && (opp & ~shaped)) return why(reason, LZ_WHY_IN_SHAPE);
        apply_rule_t<false>(s, counter ? kActCounter : kActNoMoves, a, -1);
        return LZ_OK;
    }
    default:
        return LZ_ERR_ARG;
    }
}

}  // namespace

extern "C" {

LZ_API int lz_scalar_generate(const LzScalarState* state, int what, int32_t* out, int32_t cap, int32_t* count) {
    if (!state || !out || !count || cap < 0) return LZ_ERR_ARG;
    const State s = load(*state);
    Out o{out, cap, 0, false};
    switch (what) {
    case LZ_LIST_PLACEMENT_POSITIONS:
        if (s.phase == kPlacement) o.cells(~(s.black | s.white));
        break;
    case LZ_LIST_MARK_TARGETS: o.cells(mark_targets(s)); break;
    case LZ_LIST_MOVEMENT_MOVES: movement_moves(s, o, false); break;
    case LZ_LIST_CAPTURE_TARGETS: o.cells(capture_targets(s)); break;
    case LZ_LIST_FORCED_REMOVAL_MOVES: o.moves(kForcedRemoval, kTypeForced, forced_targets(s)); break;
    case LZ_LIST_NO_MOVES_OPTIONS: o.moves(kMovement, kTypeNoMoves, removal_pool(s, kMovement)); break;
    case LZ_LIST_COUNTER_REMOVAL_MOVES: o.moves(kCounterRemoval, kTypeCounter, removal_pool(s, kCounterRemoval)); break;
    case LZ_LIST_ALL_LEGAL_MOVES:
        if (game_status(s) != 0) break;                              // IsGameOver: winner or move / no-capture limit
        switch (s.phase) {
        case kPlacement: o.moves(kPlacement, kTypePlace, ~(s.black | s.white)); break;
        case kMarkSelection: o.moves(kMarkSelection, kTypeMark, mark_targets(s)); break;
        case kRemoval: o.put(kRemoval); o.put(kTypeProcess); o.put(-1); o.put(-1); break;
        case kForcedRemoval: o.moves(kForcedRemoval, kTypeForced, forced_targets(s)); break;
        case kMovement:
            if (any_movement(s)) movement_moves(s, o, true);
            else o.moves(kMovement, kTypeNoMoves, removal_pool(s, kMovement));
            break;
        case kCaptureSelection: o.moves(kCaptureSelection, kTypeCapture, capture_targets(s)); break;
        case kCounterRemoval: o.moves(kCounterRemoval, kTypeCounter, removal_pool(s, kCounterRemoval)); break;
        default: break;
        }
        break;
    default:
        return LZ_ERR_ARG;
    }
    if (o.overflow) return LZ_ERR_ARG;
    *count = o.n;
    return LZ_OK;
}

LZ_API int lz_scalar_has_movement(const LzScalarState* state, int32_t* has_moves, int32_t* reason) {
    if (!state || !has_moves) return LZ_ERR_ARG;
    const State s = load(*state);
    if (s.phase != kMovement) return why(reason, LZ_WHY_PHASE);
    *has_moves = any_movement(s) ? 1 : 0;
    return LZ_OK;
}

LZ_API int lz_scalar_apply(const LzScalarState* state, int what, int32_t a, int32_t b, LzScalarState* next, int32_t* reason) {
    if (!state || !next) return LZ_ERR_ARG;
    if (reason) *reason = LZ_WHY_NONE;
    State s = load(*state);
    const int st = step(s, what, a, b, reason);
    if (st != LZ_OK) return st;
    store(s, *next);
    return LZ_OK;
}

LZ_API int lz_scalar_apply_move(const LzScalarState* state, const LzScalarMove* move, LzScalarState* next, int32_t* reason) {
    if (!state || !move || !next) return LZ_ERR_ARG;
    if (reason) *reason = LZ_WHY_NONE;
    State s = load(*state);
    if (move->phase != s.phase) return why(reason, LZ_WHY_MOVE_PHASE_MISMATCH);
    int what = 0;
    switch (s.phase) {
    case kPlacement: what = move->action_type == kTypePlace ? LZ_STEP_PLACEMENT : 0; break;
    case kMarkSelection: what = move->action_type == kTypeMark ? LZ_STEP_MARK : 0; break;
    case kRemoval: what = move->action_type == kTypeProcess ? LZ_STEP_PROCESS_REMOVAL : 0; break;
    case kForcedRemoval: what = move->action_type == kTypeForced ? LZ_STEP_FORCED_REMOVAL : 0; break;
    case kMovement:
        what = move->action_type == kTypeMove ? LZ_STEP_MOVEMENT : move->action_type == kTypeNoMoves ? LZ_STEP_NO_MOVES : 0;
        break;
    case kCaptureSelection: what = move->action_type == kTypeCapture ? LZ_STEP_CAPTURE : 0; break;
    case kCounterRemoval: what = move->action_type == kTypeCounter ? LZ_STEP_COUNTER_REMOVAL : 0; break;
    default: break;
    }
    if (what == 0) return why(reason, LZ_WHY_MOVE_TYPE);
    const int phase_before = s.phase, pieces_before = popc((s.black | s.white) & kFull), msc_before = s.msc;
    const int count_before = s.move_count;
    const int st = step(s, what, move->primary, move->secondary, reason);
    if (st != LZ_OK) return st;
    s.move_count = count_before + 1;
    if (phase_before == kPlacement || phase_before == kMarkSelection) s.msc = 0;
    else s.msc = popc((s.black | s.white) & kFull) < pieces_before ? 0 : msc_before + 1;
    store(s, *next);
    return LZ_OK;
}

LZ_API int lz_scalar_status(const LzScalarState* state, int32_t* winner, int32_t* game_over) {
    if (!state || !winner || !game_over) return LZ_ERR_ARG;
    const int g = game_status(load(*state));
    *winner = (g == 1 || g == -1) ? g : 0;
    *game_over = g != 0 ? 1 : 0;
    return LZ_OK;
}

LZ_API int lz_scalar_piece_in_shape(const LzScalarState* state, int32_t cell, int32_t player, int32_t use_marks, int32_t* in_shape) {
    if (!state || !in_shape || (player != 1 && player != -1)) return LZ_ERR_ARG;
    *in_shape = 0;
    if (!on_board(cell)) return LZ_OK;
    const State s = load(*state);
    const uint64_t P = side(s, player), m = use_marks ? marks_of(s, player) : 0;
    *in_shape = (in_shape_set(P, P & ~m) >> cell) & 1 ? 1 : 0;
    return LZ_OK;
}

}  // extern "C"
