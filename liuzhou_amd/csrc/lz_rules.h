// lz_rules.h -- Liuzhou rule engine on 36-bit bitboards (host + device).
//
// One game state = 4 bitboards (black, white, marks_black, marks_white; bit i = cell r*6+c) plus a
// few small counters.  Everything the reference does with 36-cell scans (Fang 2x2 / Zhou full-line
// detection, target pools, legal moves, transitions) becomes a handful of shifts/ands here, so a
// single GPU lane can carry a whole state in registers.
//
// Behaviour follows (paths relative to the reference repo):
//   src/rule_engine.py:22-551, src/move_generator.py:24-139          (rules + transitions)
//   v0/src/game/fast_legal_mask.cpp:110-418                           (tensor-mask semantics,
//                                                                       incl. unreachable states)
//   v0/src/game/fast_apply_moves_cuda.cu:240-744                      (illegal action == no-op)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LZ_HD __host__ __device__ __forceinline__
#else
#define LZ_HD inline
#endif

namespace lz {

constexpr int kCells = 36;
constexpr uint64_t kFull = 0xFFFFFFFFFull;          // 36 ones
constexpr uint64_t kCol0 = 0x041041041ull;          // bit 6r for r=0..5
constexpr uint64_t kCol5 = kCol0 << 5;
constexpr uint64_t kAnchor = 0x1F7DF7DFull;         // rows 0..4, cols 0..4 (2x2 top-left corners)
constexpr int kMaxMoveCount = 144;                  // src/game_state.py:29
constexpr int kLoseThreshold = 4;                   // src/game_state.py:30
constexpr int kNoCaptureLimit = 36;                 // src/game_state.py:31

enum Phase : int { kPlacement = 1, kMarkSelection = 2, kRemoval = 3, kMovement = 4,
                   kCaptureSelection = 5, kForcedRemoval = 6, kCounterRemoval = 7 };
// v0/src/game/fast_legal_mask_common.hpp:33-43
enum ActionKind : int { kActInvalid = 0, kActPlace = 1, kActMove = 2, kActMark = 3, kActCapture = 4,
                        kActForced = 5, kActCounter = 6, kActNoMoves = 7, kActProcess = 8 };

LZ_HD int popc(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
LZ_HD int ctz(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}

struct State {
    uint64_t black, white, mb, mw;
    int phase, player;
    int pm_req, pm_rem, pc_req, pc_rem;
    int forced, move_count, msc;
};

// ---- packed 32-byte form used by the device-resident engine -------------------------------------
// w0 = black | move_count<<36 | msc<<44 | phase<<50 | (player==-1)<<53 | forced<<54
//            | pm_req<<56 | pm_rem<<58 | pc_req<<60 | pc_rem<<62
struct Packed { uint64_t w0, w1, w2, w3; };

LZ_HD Packed pack(const State& s) {
    Packed p;
    p.w0 = (s.black & kFull) | ((uint64_t)(s.move_count & 0xFF) << 36) | ((uint64_t)(s.msc & 0x3F) << 44) |
           ((uint64_t)(s.phase & 7) << 50) | ((uint64_t)(s.player < 0 ? 1 : 0) << 53) |
           ((uint64_t)(s.forced & 3) << 54) | ((uint64_t)(s.pm_req & 3) << 56) |
           ((uint64_t)(s.pm_rem & 3) << 58) | ((uint64_t)(s.pc_req & 3) << 60) | ((uint64_t)(s.pc_rem & 3) << 62);
    p.w1 = s.white & kFull; p.w2 = s.mb & kFull; p.w3 = s.mw & kFull;
    return p;
}
LZ_HD State unpack(const Packed& p) {
    State s;
    s.black = p.w0 & kFull; s.white = p.w1 & kFull; s.mb = p.w2 & kFull; s.mw = p.w3 & kFull;
    s.move_count = (int)((p.w0 >> 36) & 0xFF); s.msc = (int)((p.w0 >> 44) & 0x3F);
    s.phase = (int)((p.w0 >> 50) & 7); s.player = ((p.w0 >> 53) & 1) ? -1 : 1;
    s.forced = (int)((p.w0 >> 54) & 3); s.pm_req = (int)((p.w0 >> 56) & 3); s.pm_rem = (int)((p.w0 >> 58) & 3);
    s.pc_req = (int)((p.w0 >> 60) & 3); s.pc_rem = (int)((p.w0 >> 62) & 3);
    return s;
}

// ---- shape detection ----------------------------------------------------------------------------
// cells that are a corner of a 2x2 block fully inside U              (rule_engine.py:482-499)
LZ_HD uint64_t squares_of(uint64_t U) {
    uint64_t f = U & (U >> 1) & (U >> 6) & (U >> 7) & kAnchor;
    return (f | (f << 1) | (f << 6) | (f << 7)) & kFull;
}

// Pieces of P that are "in a shape" given the unmarked subset U = P & ~marked.
// check_lines (rule_engine.py:502-538) counts the probed cell itself WITHOUT looking at its own
// mark and every other cell of the line only if unmarked; so besides full lines of U, a line of U
// that misses exactly one cell which is a (marked) piece of P puts that one cell in shape.
LZ_HD uint64_t in_shape_set(uint64_t P, uint64_t U) {
    uint64_t res = squares_of(U);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        uint64_t line = 0x3Full << (6 * r);
        uint64_t got = U & line;
        if (got == line) res |= line;
        else if (popc(got) == 5) res |= (line & ~got) & P;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        uint64_t line = kCol0 << c;
        uint64_t got = U & line;
        if (got == line) res |= line;
        else if (popc(got) == 5) res |= (line & ~got) & P;
    }
    return res & P;
}

// detect_shape_formed for a freshly placed / moved piece at `cell` (cell is in U): 2 Zhou, 1 Fang, 0
// (rule_engine.py:465-479, Zhou beats Fang)
LZ_HD int detect_shape(uint64_t U, int cell) {
    int r = cell / 6, c = cell - 6 * r;
    uint64_t row = 0x3Full << (6 * r), col = kCol0 << c;
    if ((U & row) == row || (U & col) == col) return 2;
    return (squares_of(U) >> cell) & 1 ? 1 : 0;
}

// prefer_normal_pieces (fast_legal_mask.cpp:110-132)
LZ_HD uint64_t prefer_normal(uint64_t cand, uint64_t P, uint64_t U) {
    uint64_t normal = cand & ~in_shape_set(P, U);
    return normal ? normal : cand;
}

// ---- legal actions ------------------------------------------------------------------------------
struct Legal {
    uint64_t place;        // bit = empty cell (phase 1)
    uint64_t up, down, left, right;   // from-cells that may move in each direction (phase 4)
    uint64_t sel;          // selection targets
    int sel_kind;          // ActionKind of the selection entries (0 if none)
    int process;           // 1 iff phase == removal
};

// own / opp / empty are the cells equal to +player, -player and 0 (the SoA op passes them from the
// raw bytes so that unreachable inputs behave like the reference's per-cell compares).
// fallback_forced = 1: tensor semantics (fast_legal_mask.cpp:134-151); 0: python (move_generator.py:147-172)
LZ_HD Legal legal_actions(uint64_t black, uint64_t white, uint64_t own, uint64_t opp, uint64_t empty,
                          uint64_t mb, uint64_t mw, int phase, int player, int pm_rem, int pc_rem,
                          int forced, int fallback_forced) {
    Legal L;
    L.place = 0; L.up = L.down = L.left = L.right = 0; L.sel = 0; L.sel_kind = 0; L.process = 0;
    if (phase == kPlacement) L.place = empty & kFull;
    bool has_movement = false;
    if (phase == kMovement) {
        L.up = own & (empty << 6) & kFull;                  // up:    dest = from - 6
        L.down = own & (empty >> 6) & kFull;                // down:  dest = from + 6
        L.left = own & (empty << 1) & ~kCol0 & kFull;       // left:  dest = from - 1, col >= 1
        L.right = own & (empty >> 1) & ~kCol5 & kFull;      // right: dest = from + 1, col <= 4
        has_movement = (L.up | L.down | L.left | L.right) != 0;
    }
    if (phase == kMarkSelection) {
        uint64_t om = (player == 1) ? mw : mb;              // fast_legal_mask.cpp:360
        L.sel_kind = kActMark;
        if (pm_rem > 0) L.sel = prefer_normal(opp & ~om, opp, opp & ~om);
    } else if (phase == kCaptureSelection) {
        uint64_t om = (player == 1) ? mw : mb;
        L.sel_kind = kActCapture;
        if (pc_rem > 0) L.sel = prefer_normal(opp, opp, opp & ~om);   // candidacy ignores marks
    } else if (phase == kForcedRemoval) {
        L.sel_kind = kActForced;
        if (forced < 2) {
            uint64_t tgt = (forced == 0) ? black : white;
            L.sel = fallback_forced ? prefer_normal(tgt, tgt, tgt) : (tgt & ~in_shape_set(tgt, tgt));
        }
    } else if (phase == kCounterRemoval) {
        L.sel_kind = kActCounter;
        L.sel = prefer_normal(opp, opp, opp);
    } else if (phase == kMovement && !has_movement) {
        L.sel_kind = kActNoMoves;
        L.sel = prefer_normal(opp, opp, opp);
    }
    if (phase == kRemoval) L.process = 1;
    return L;
}

LZ_HD Legal legal_actions(const State& s, int fallback_forced) {
    uint64_t own = s.player == 1 ? s.black : s.white;
    uint64_t opp = s.player == 1 ? s.white : s.black;
    uint64_t empty = ~(s.black | s.white) & kFull;
    return legal_actions(s.black, s.white, own, opp, empty, s.mb, s.mw, s.phase, s.player, s.pm_rem,
                         s.pc_rem, s.forced, fallback_forced);
}

// 220-d index helpers (v0/python/move_encoder.py:46-51: 36 / 144 / 36 / 4)
LZ_HD uint64_t move_set(const Legal& L, int d) {   // selects, not an indexed array (stays in registers)
    return d == 0 ? L.up : d == 1 ? L.down : d == 2 ? L.left : L.right;
}
LZ_HD bool legal_bit(const Legal& L, int a) {
    if (a < 36) return (L.place >> a) & 1;
    if (a < 180) { int from = (a - 36) >> 2, d = (a - 36) & 3; return (move_set(L, d) >> from) & 1; }
    if (a < 216) return (L.sel >> (a - 180)) & 1;
    return a == 216 && L.process;
}
LZ_HD int legal_count(const Legal& L) {
    return popc(L.place) + popc(L.up) + popc(L.down) + popc(L.left) + popc(L.right) + popc(L.sel) + L.process;
}
LZ_HD int move_dest(int from, int dir) {
    return dir == 0 ? from - 6 : dir == 1 ? from + 6 : dir == 2 ? from - 1 : from + 1;
}

// game status: 0 running, +1 black wins, -1 white wins, 2 draw       (game_state.py:87-96,165-181)
LZ_HD int game_status(const State& s) {
    if (s.phase == kMovement || s.phase == kCaptureSelection || s.phase == kCounterRemoval) {
        if (popc(s.black) < kLoseThreshold) return -1;
        if (popc(s.white) < kLoseThreshold) return 1;
    }
    if (s.move_count >= kMaxMoveCount || s.msc >= kNoCaptureLimit) return 2;
    return 0;
}

// ---- transitions --------------------------------------------------------------------------------
// Illegal actions leave the state untouched (CUDA semantics, fast_apply_moves_cuda.cu:240-546);
// returns true iff the action was accepted.  move_count / moves_since_capture are handled by apply().
// CHECK = false: the caller guarantees a legal action (the tree engine only applies actions it enumerated with
// legal_actions): the validation -- and the shape sets computed only for it -- is compiled out; results are the same.
template <bool CHECK>
LZ_HD bool apply_rule_t(State& s, int kind, int primary, int secondary) {
    const uint64_t full = ((s.black | s.white) & kFull);
    switch (kind) {
    case kActPlace: {
        if (CHECK && (s.phase != kPlacement || primary < 0 || primary >= kCells)) return false;
        uint64_t bit = 1ull << primary;
        if (CHECK && (full & bit)) return false;
        uint64_t om = s.player == 1 ? s.mw : s.mb;
        if (CHECK && (om & bit)) return false;
        uint64_t& own = s.player == 1 ? s.black : s.white;
        own |= bit;
        uint64_t ownm = s.player == 1 ? s.mb : s.mw;
        if (!(ownm & bit)) {
            int shape = detect_shape(own & ~ownm, primary);
            if (shape) { s.pm_req = s.pm_rem = shape; s.phase = kMarkSelection; return true; }
        }
        s.pm_req = s.pm_rem = 0;
        if (((s.black | s.white) & kFull) == kFull) s.phase = kRemoval;
        else { s.player = -s.player; s.phase = kPlacement; }
        return true;
    }
    case kActMark: {
        if (CHECK && (s.phase != kMarkSelection || s.pm_rem <= 0 || primary < 0 || primary >= kCells)) return false;
        uint64_t bit = 1ull << primary;
        uint64_t opp = s.player == 1 ? s.white : s.black;
        uint64_t& om = s.player == 1 ? s.mw : s.mb;
        if (CHECK && (!(opp & bit) || (om & bit))) return false;
        uint64_t U = opp & ~om;
        uint64_t shp = in_shape_set(opp, U);
        if (CHECK && ((shp & bit) && (U & ~shp))) return false;     // unmarked normal pieces remain
        om |= bit;
        s.pm_rem -= 1;
        if (s.pm_rem > 0) return true;
        s.pm_req = s.pm_rem = 0;
        if (full == kFull) s.phase = kRemoval;
        else { s.player = -s.player; s.phase = kPlacement; }
        return true;
    }
    case kActProcess: {
        if (CHECK && (s.phase != kRemoval)) return false;
        uint64_t m = (s.mb | s.mw) & kFull;
        if (!m) { s.phase = kForcedRemoval; s.player = -1; s.forced = 0; return true; }
        s.black &= ~m; s.white &= ~m; s.mb = 0; s.mw = 0;
        s.phase = kMovement; s.player = -1;
        return true;
    }
    case kActForced: {
        if (CHECK && (s.phase != kForcedRemoval || primary < 0 || primary >= kCells)) return false;
        uint64_t bit = 1ull << primary;
        if (s.forced == 0) {
            if (CHECK && (s.player != -1 || !(s.black & bit))) return false;
            if (CHECK && (in_shape_set(s.black, s.black) & bit)) return false;
            s.black &= ~bit; s.forced = 1; s.player = 1;
            return true;
        }
        if (s.forced == 1) {
            if (CHECK && (s.player != 1 || !(s.white & bit))) return false;
            if (CHECK && (in_shape_set(s.white, s.white) & bit)) return false;
            s.white &= ~bit; s.forced = 2; s.phase = kMovement; s.player = -1;
            return true;
        }
        return false;
    }
    case kActMove: {
        if (CHECK && (s.phase != kMovement || secondary < 0 || secondary >= 4 || primary < 0 || primary >= kCells)) return false;
        int r = primary / 6, c = primary - 6 * r;
        if (CHECK && ((secondary == 0 && r == 0) || (secondary == 1 && r == 5) || (secondary == 2 && c == 0) ||
            (secondary == 3 && c == 5))) return false;
        int to = move_dest(primary, secondary);
        uint64_t fb = 1ull << primary, tb = 1ull << to;
        uint64_t& own = s.player == 1 ? s.black : s.white;
        if (CHECK && (!(own & fb) || (full & tb))) return false;
        own = (own & ~fb) | tb;
        int shape = detect_shape(own, to);
        if (shape) { s.pc_req = s.pc_rem = shape; s.phase = kCaptureSelection; return true; }
        s.pc_req = s.pc_rem = 0;
        s.player = -s.player;
        return true;
    }
    case kActNoMoves: {
        if (CHECK && (s.phase != kMovement || primary < 0 || primary >= kCells)) return false;
        uint64_t bit = 1ull << primary;
        uint64_t& opp = s.player == 1 ? s.white : s.black;
        if (CHECK && (!(opp & bit))) return false;
        uint64_t shp = in_shape_set(opp, opp);
        if (CHECK && ((shp & bit) && (opp & ~shp))) return false;
        opp &= ~bit;
        if (popc(opp) < kLoseThreshold) return true;
        s.phase = kCounterRemoval; s.player = -s.player;
        return true;
    }
    case kActCapture: {
        if (CHECK && (s.phase != kCaptureSelection || s.pc_rem <= 0 || primary < 0 || primary >= kCells)) return false;
        uint64_t bit = 1ull << primary;
        uint64_t& opp = s.player == 1 ? s.white : s.black;
        uint64_t om = s.player == 1 ? s.mw : s.mb;
        if (CHECK && (!(opp & bit))) return false;
        uint64_t shp = in_shape_set(opp, opp & ~om);
        if (CHECK && ((shp & bit) && (opp & ~shp))) return false;
        opp &= ~bit;
        s.pc_rem -= 1;
        if (popc(opp) < kLoseThreshold || s.pc_rem > 0) return true;
        s.pc_req = s.pc_rem = 0;
        s.player = -s.player; s.phase = kMovement;
        return true;
    }
    case kActCounter: {
        if (CHECK && (s.phase != kCounterRemoval || primary < 0 || primary >= kCells)) return false;
        uint64_t bit = 1ull << primary;
        uint64_t& stuck = s.player == 1 ? s.white : s.black;
        if (CHECK && (!(stuck & bit))) return false;
        uint64_t shp = in_shape_set(stuck, stuck);
        if (CHECK && ((shp & bit) && (stuck & ~shp))) return false;
        stuck &= ~bit;
        if (popc(stuck) < kLoseThreshold) return true;
        s.phase = kMovement; s.player = -s.player;
        return true;
    }
    default: return false;
    }
}
LZ_HD bool apply_rule(State& s, int kind, int primary, int secondary) { return apply_rule_t<true>(s, kind, primary, secondary); }

// apply_action incl. move_count / moves_since_capture bookkeeping
// (fast_apply_moves_cuda.cu:610-743: placement bumps move_count only when accepted, every other
//  known kind always; unknown kinds touch nothing but moves_since_capture; move_generator.py:122-137)
template <bool CHECK>
LZ_HD bool apply_t(State& s, int kind, int primary, int secondary) {
    const int phase_before = s.phase;
    const int old_total = popc((s.black | s.white) & kFull);
    const int old_msc = s.msc;
    bool ok = apply_rule_t<CHECK>(s, kind, primary, secondary);
    if (kind == kActPlace) { if (ok) s.move_count += 1; }
    else if (kind >= kActMove && kind <= kActProcess) s.move_count += 1;
    if (phase_before == kPlacement || phase_before == kMarkSelection) s.msc = 0;
    else s.msc = (popc((s.black | s.white) & kFull) < old_total) ? 0 : old_msc + 1;
    return ok;
}
LZ_HD bool apply(State& s, int kind, int primary, int secondary) { return apply_t<true>(s, kind, primary, secondary); }
// a legal action of `s` (enumerated by legal_actions with Python semantics): same result, no validation
LZ_HD void apply_legal(State& s, int kind, int primary, int secondary) { (void)apply_t<false>(s, kind, primary, secondary); }

// 220-d action index -> (kind, primary, secondary, extra) for the state's phase
// (v0/python/move_encoder.py:164-247; metadata layout fast_legal_mask.cpp:323-345)
LZ_HD void index_to_code(int phase, int a, int& kind, int& primary, int& secondary, int& extra) {
    kind = kActInvalid; primary = secondary = extra = -1;
    if (a < 0) return;
    if (a < 36) { kind = kActPlace; primary = a; return; }
    if (a < 180) { kind = kActMove; primary = (a - 36) >> 2; secondary = (a - 36) & 3; extra = move_dest(primary, secondary); return; }
    if (a < 216) {
        primary = a - 180;
        kind = phase == kMarkSelection ? kActMark : phase == kCaptureSelection ? kActCapture :
               phase == kForcedRemoval ? kActForced : phase == kCounterRemoval ? kActCounter :
               phase == kMovement ? kActNoMoves : kActInvalid;
        return;
    }
    if (a == 216) kind = kActProcess;
}

}  // namespace lz
