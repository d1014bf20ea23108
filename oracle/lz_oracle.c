/*
 * lz_oracle.c -- CPU ORACLE (test infrastructure, see lz_oracle.h).
 *
 * Plain, array-based, deliberately *not* bitboard-based: it is an independent restatement of
 * the reference so that the bitboard HIP kernels are checked against a different formulation.
 * All reference citations are relative to /root/reference.
 */
#include "lz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { PH_PLACEMENT = 1, PH_MARK = 2, PH_REMOVAL = 3, PH_MOVEMENT = 4, PH_CAPTURE = 5,
       PH_FORCED = 6, PH_COUNTER = 7 };
/* v0/src/game/fast_legal_mask_common.hpp:33-43 */
enum { K_INVALID = 0, K_PLACE = 1, K_MOVE = 2, K_MARK = 3, K_CAPTURE = 4, K_FORCED = 5,
       K_COUNTER = 6, K_NOMOVES = 7, K_PROCESS = 8 };
#define SIZE 6
#define MAX_MOVE_COUNT 144      /* src/game_state.py:29 */
#define LOSE_THRESHOLD 4        /* src/game_state.py:30 */
#define NO_CAPTURE_LIMIT 36     /* src/game_state.py:31 */
static const int DR[4] = {-1, 1, 0, 0};  /* src/rule_engine.py:207: up, down, left, right */
static const int DC[4] = {0, 0, -1, 1};

/* ------------------------------------------------------------------------------------------
 * shape detection: src/rule_engine.py:482-551 (check_squares / check_lines / is_piece_in_shape)
 * `marked` may be NULL (== empty set).
 * ---------------------------------------------------------------------------------------- */
static int is_marked(const uint8_t* marked, int idx) { return marked != NULL && marked[idx] != 0; }

static int check_squares(const int8_t* b, const uint8_t* marked, int r, int c, int pv) {
    static const int off[2] = {0, -1};
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 2; ++j) {
            int rr = r + off[i], cc = c + off[j];
            if (rr >= 0 && rr < SIZE - 1 && cc >= 0 && cc < SIZE - 1) {
                const int cells[4] = {rr * SIZE + cc, rr * SIZE + cc + 1,
                                      (rr + 1) * SIZE + cc, (rr + 1) * SIZE + cc + 1};
                int ok = 1;
                for (int k = 0; k < 4; ++k) {
                    if (b[cells[k]] != pv || is_marked(marked, cells[k])) { ok = 0; break; }
                }
                if (ok) return 1;
            }
        }
    }
    return 0;
}

static int check_lines(const int8_t* b, const uint8_t* marked, int r, int c, int pv) {
    /* NB: the cell itself counts 1 without looking at its own mark (rule_engine.py:513) */
    int count = 1;
    for (int dc = c - 1; dc >= 0; --dc) {
        int idx = r * SIZE + dc;
        if (b[idx] == pv && !is_marked(marked, idx)) ++count; else break;
    }
    for (int dc = c + 1; dc < SIZE; ++dc) {
        int idx = r * SIZE + dc;
        if (b[idx] == pv && !is_marked(marked, idx)) ++count; else break;
    }
    if (count >= 6) return 1;
    count = 1;
    for (int dr = r - 1; dr >= 0; --dr) {
        int idx = dr * SIZE + c;
        if (b[idx] == pv && !is_marked(marked, idx)) ++count; else break;
    }
    for (int dr = r + 1; dr < SIZE; ++dr) {
        int idx = dr * SIZE + c;
        if (b[idx] == pv && !is_marked(marked, idx)) ++count; else break;
    }
    return count >= 6;
}

static int in_shape(const int8_t* b, const uint8_t* marked, int cell, int pv) {
    if (b[cell] != pv) return 0;
    int r = cell / SIZE, c = cell % SIZE;
    return check_squares(b, marked, r, c, pv) || check_lines(b, marked, r, c, pv);
}

/* src/rule_engine.py:465-479: 2 = line (Zhou wins), 1 = square, 0 = none */
static int detect_shape(const int8_t* b, const uint8_t* marked, int cell, int pv) {
    int r = cell / SIZE, c = cell % SIZE;
    int sq = check_squares(b, marked, r, c, pv);
    int ln = check_lines(b, marked, r, c, pv);
    if (ln) return 2;
    if (sq) return 1;
    return 0;
}

static int count_pieces(const int8_t* b, int pv) {
    int n = 0;
    for (int i = 0; i < LZO_CELLS; ++i) n += (b[i] == pv);
    return n;
}

static int board_full(const int8_t* b) {
    for (int i = 0; i < LZO_CELLS; ++i) if (b[i] == 0) return 0;
    return 1;
}

/* fast_legal_mask.cpp:110-132: keep non-shape candidates, else all candidates */
static int prefer_normal(const int8_t* b, const uint8_t* marked, const int* cand, int n, int pv,
                         int* out) {
    int m = 0;
    for (int i = 0; i < n; ++i) if (!in_shape(b, marked, cand[i], pv)) out[m++] = cand[i];
    if (m > 0) return m;
    for (int i = 0; i < n; ++i) out[i] = cand[i];
    return n;
}

/* ------------------------------------------------------------------------------------------
 * SoA <-> scalar state
 * ---------------------------------------------------------------------------------------- */
static void load_state(const lzo_batch* in, int64_t b, lzo_state* s) {
    memcpy(s->board, in->board + b * LZO_CELLS, LZO_CELLS);
    memcpy(s->mb, in->marks_black + b * LZO_CELLS, LZO_CELLS);
    memcpy(s->mw, in->marks_white + b * LZO_CELLS, LZO_CELLS);
    s->phase = in->phase[b];
    s->player = in->current_player[b];
    s->pm_req = in->pending_marks_required[b];
    s->pm_rem = in->pending_marks_remaining[b];
    s->pc_req = in->pending_captures_required[b];
    s->pc_rem = in->pending_captures_remaining[b];
    s->forced = in->forced_removals_done[b];
    s->move_count = in->move_count[b];
    s->msc = in->moves_since_capture[b];
}

static void store_state(lzo_batch* out, int64_t i, const lzo_state* s) {
    memcpy(out->board + i * LZO_CELLS, s->board, LZO_CELLS);
    memcpy(out->marks_black + i * LZO_CELLS, s->mb, LZO_CELLS);
    memcpy(out->marks_white + i * LZO_CELLS, s->mw, LZO_CELLS);
    out->phase[i] = s->phase;
    out->current_player[i] = s->player;
    out->pending_marks_required[i] = s->pm_req;
    out->pending_marks_remaining[i] = s->pm_rem;
    out->pending_captures_required[i] = s->pc_req;
    out->pending_captures_remaining[i] = s->pc_rem;
    out->forced_removals_done[i] = s->forced;
    out->move_count[i] = s->move_count;
    out->moves_since_capture[i] = s->msc;
}

/* ------------------------------------------------------------------------------------------
 * selection target pools (tensor semantics): fast_legal_mask.cpp:134-251
 * returns count, kind in *kind (0 if this phase has no selection pool)
 * ---------------------------------------------------------------------------------------- */
static int selection_targets(const lzo_state* s, int has_movement, int fallback_forced,
                             int* out, int* kind) {
    const int8_t* b = s->board;
    int cur = (int)s->player;
    int cand[LZO_CELLS];
    int n = 0;
    *kind = 0;
    if (s->phase == PH_MARK) {
        /* fast_legal_mask.cpp:203-226 */
        const uint8_t* opp_marked = (cur == 1) ? s->mw : s->mb;
        int opp = -cur;
        *kind = K_MARK;
        for (int i = 0; i < LZO_CELLS; ++i)
            if (b[i] == opp && !is_marked(opp_marked, i)) cand[n++] = i;
        int m = prefer_normal(b, opp_marked, cand, n, opp, out);
        return s->pm_rem > 0 ? m : 0;
    }
    if (s->phase == PH_CAPTURE) {
        /* fast_legal_mask.cpp:228-251: candidacy ignores marks, shape test uses them */
        const uint8_t* opp_marked = (cur == 1) ? s->mw : s->mb;
        int opp = -cur;
        *kind = K_CAPTURE;
        for (int i = 0; i < LZO_CELLS; ++i) if (b[i] == opp) cand[n++] = i;
        if (s->pc_rem > 0) return prefer_normal(b, opp_marked, cand, n, opp, out);
        return 0;
    }
    if (s->phase == PH_FORCED) {
        /* fast_legal_mask.cpp:134-151 */
        *kind = K_FORCED;
        if (s->forced >= 2) return 0;
        int v = (s->forced == 0) ? 1 : -1;
        for (int i = 0; i < LZO_CELLS; ++i) if (b[i] == v) cand[n++] = i;
        if (fallback_forced) return prefer_normal(b, NULL, cand, n, v, out);
        /* python: src/move_generator.py:147-172 -- no fallback */
        int m = 0;
        for (int i = 0; i < n; ++i) if (!in_shape(b, NULL, cand[i], v)) out[m++] = cand[i];
        return m;
    }
    if (s->phase == PH_COUNTER || (s->phase == PH_MOVEMENT && !has_movement)) {
        /* fast_legal_mask.cpp:153-183 */
        int v = -cur;
        *kind = (s->phase == PH_COUNTER) ? K_COUNTER : K_NOMOVES;
        for (int i = 0; i < LZO_CELLS; ++i) if (b[i] == v) cand[n++] = i;
        return prefer_normal(b, NULL, cand, n, v, out);
    }
    return 0;
}

/* v0/src/game/fast_legal_mask.cpp:253-418 */
void lzo_encode_actions(const lzo_batch* in, int64_t B, int64_t pd, int64_t md, int64_t sd,
                        int64_t ad, uint8_t* mask, int32_t* meta) {
    const int64_t T = pd + md + sd + ad;
    memset(mask, 0, (size_t)(B * T));
    for (int64_t i = 0; i < B * T * LZO_META; ++i) meta[i] = -1;
    for (int64_t bi = 0; bi < B; ++bi) {
        lzo_state s;
        load_state(in, bi, &s);
        uint8_t* mrow = mask + bi * T;
        int32_t* trow = meta + bi * T * LZO_META;
        int cur = (int)s.player;
#define SET_META(g, k, p, q, e) do { mrow[g] = 1; trow[(g)*4+0] = (k); trow[(g)*4+1] = (p); \
                                     trow[(g)*4+2] = (q); trow[(g)*4+3] = (e); } while (0)
        if (s.phase == PH_PLACEMENT) {
            for (int c = 0; c < LZO_CELLS; ++c) if (s.board[c] == 0) SET_META(c, K_PLACE, c, -1, -1);
        }
        int has_movement = 0;
        if (s.phase == PH_MOVEMENT) {
            for (int c = 0; c < LZO_CELLS; ++c) {
                if (s.board[c] != cur) continue;
                int r = c / SIZE, cc = c % SIZE;
                for (int d = 0; d < 4; ++d) {
                    int nr = r + DR[d], nc = cc + DC[d];
                    if (nr < 0 || nr >= SIZE || nc < 0 || nc >= SIZE) continue;
                    int dest = nr * SIZE + nc;
                    if (s.board[dest] == 0) {
                        int64_t g = pd + c * 4 + d;
                        SET_META(g, K_MOVE, c, d, dest);
                        has_movement = 1;
                    }
                }
            }
        }
        int targets[LZO_CELLS], kind = 0;
        int n = selection_targets(&s, has_movement, /*fallback_forced=*/1, targets, &kind);
        for (int i = 0; i < n; ++i) {
            int idx = targets[i];
            if (idx >= 0 && idx < sd) { int64_t g = pd + md + idx; SET_META(g, kind, idx, -1, -1); }
        }
        if (s.phase == PH_REMOVAL && ad > 0) { int64_t g = pd + md + sd; SET_META(g, K_PROCESS, -1, -1, -1); }
#undef SET_META
    }
}

/* ------------------------------------------------------------------------------------------
 * transitions: src/rule_engine.py:22-457 == fast_apply_moves.cpp:246-593
 * each returns 0 = applied, -1 = illegal (state untouched)
 * ---------------------------------------------------------------------------------------- */
static int has_unmarked_normal(const int8_t* b, const uint8_t* marked, int pv) {
    for (int i = 0; i < LZO_CELLS; ++i)
        if (b[i] == pv && !in_shape(b, marked, i, pv) && !is_marked(marked, i)) return 1;
    return 0;
}
static int has_normal(const int8_t* b, const uint8_t* marked, int pv) {
    for (int i = 0; i < LZO_CELLS; ++i) if (b[i] == pv && !in_shape(b, marked, i, pv)) return 1;
    return 0;
}

/* rule_engine.py:22-74; note fast_apply_moves.cpp:246-311 bumps move_count inside placement */
static int do_placement(lzo_state* s, int cell) {
    if (s->phase != PH_PLACEMENT) return -1;
    if (cell < 0 || cell >= LZO_CELLS) return -1;
    if (s->board[cell] != 0) return -1;
    int cur = (int)s->player;
    const uint8_t* opp_marked = (cur == 1) ? s->mw : s->mb;
    if (is_marked(opp_marked, cell)) return -1;
    s->board[cell] = (int8_t)cur;
    const uint8_t* own_marked = (cur == 1) ? s->mb : s->mw;
    if (!is_marked(own_marked, cell)) {
        int shape = detect_shape(s->board, own_marked, cell, cur);
        if (shape == 2) { s->pm_req = s->pm_rem = 2; s->phase = PH_MARK; return 0; }
        if (shape == 1) { s->pm_req = s->pm_rem = 1; s->phase = PH_MARK; return 0; }
    }
    s->pm_req = s->pm_rem = 0;
    if (board_full(s->board)) s->phase = PH_REMOVAL;
    else { s->player = -s->player; s->phase = PH_PLACEMENT; }
    return 0;
}

/* rule_engine.py:109-161 */
static int do_mark(lzo_state* s, int cell) {
    if (s->phase != PH_MARK || s->pm_rem <= 0) return -1;
    if (cell < 0 || cell >= LZO_CELLS) return -1;
    int opp = (int)-s->player;
    uint8_t* opp_marked = (opp == -1) ? s->mw : s->mb;
    if (s->board[cell] != opp || opp_marked[cell]) return -1;
    if (in_shape(s->board, opp_marked, cell, opp) && has_unmarked_normal(s->board, opp_marked, opp))
        return -1;
    opp_marked[cell] = 1;
    s->pm_rem -= 1;
    if (s->pm_rem > 0) return 0;
    s->pm_req = s->pm_rem = 0;
    if (board_full(s->board)) s->phase = PH_REMOVAL;
    else { s->player = -s->player; s->phase = PH_PLACEMENT; }
    return 0;
}

/* rule_engine.py:164-193 */
static int do_process_removal(lzo_state* s) {
    if (s->phase != PH_REMOVAL) return -1;
    int any = 0;
    for (int i = 0; i < LZO_CELLS; ++i) any |= (s->mb[i] || s->mw[i]);
    if (!any) { s->phase = PH_FORCED; s->player = -1; s->forced = 0; return 0; }
    int removed = 0;
    for (int i = 0; i < LZO_CELLS; ++i) {
        if (s->mb[i]) { s->board[i] = 0; ++removed; }
        else if (s->mw[i]) { s->board[i] = 0; ++removed; }
    }
    memset(s->mb, 0, LZO_CELLS);
    memset(s->mw, 0, LZO_CELLS);
    if (removed > 0) { s->phase = PH_MOVEMENT; s->player = -1; }
    return 0;
}

/* rule_engine.py:341-372 */
static int do_forced(lzo_state* s, int cell) {
    if (s->phase != PH_FORCED || cell < 0 || cell >= LZO_CELLS) return -1;
    if (s->forced == 0) {
        if (s->player != -1 || s->board[cell] != 1) return -1;
        if (in_shape(s->board, NULL, cell, 1)) return -1;
        s->board[cell] = 0; s->forced = 1; s->player = 1;
        return 0;
    }
    if (s->forced == 1) {
        if (s->player != 1 || s->board[cell] != -1) return -1;
        if (in_shape(s->board, NULL, cell, -1)) return -1;
        s->board[cell] = 0; s->forced = 2; s->phase = PH_MOVEMENT; s->player = -1;
        return 0;
    }
    return -1;
}

/* rule_engine.py:220-264 */
static int do_movement(lzo_state* s, int from, int dir) {
    if (s->phase != PH_MOVEMENT || dir < 0 || dir >= 4) return -1;
    if (from < 0 || from >= LZO_CELLS) return -1;
    int r = from / SIZE, c = from % SIZE;
    int nr = r + DR[dir], nc = c + DC[dir];
    if (nr < 0 || nr >= SIZE || nc < 0 || nc >= SIZE) return -1;
    int to = nr * SIZE + nc;
    if (s->board[from] != s->player || s->board[to] != 0) return -1;
    s->board[to] = s->board[from];
    s->board[from] = 0;
    int shape = detect_shape(s->board, NULL, to, (int)s->player);
    if (shape == 2) { s->pc_req = s->pc_rem = 2; s->phase = PH_CAPTURE; return 0; }
    if (shape == 1) { s->pc_req = s->pc_rem = 1; s->phase = PH_CAPTURE; return 0; }
    s->pc_req = s->pc_rem = 0;
    s->player = -s->player;
    return 0;
}

/* rule_engine.py:375-414 */
static int do_no_moves(lzo_state* s, int cell) {
    if (s->phase != PH_MOVEMENT || cell < 0 || cell >= LZO_CELLS) return -1;
    int opp = (int)-s->player;
    if (s->board[cell] != opp) return -1;
    if (in_shape(s->board, NULL, cell, opp) && has_normal(s->board, NULL, opp)) return -1;
    s->board[cell] = 0;
    if (count_pieces(s->board, opp) < LOSE_THRESHOLD) return 0;
    s->phase = PH_COUNTER;
    s->player = -s->player;
    return 0;
}

/* rule_engine.py:291-338 */
static int do_capture(lzo_state* s, int cell) {
    if (s->phase != PH_CAPTURE || s->pc_rem <= 0 || cell < 0 || cell >= LZO_CELLS) return -1;
    int opp = (int)-s->player;
    const uint8_t* opp_marked = (opp == -1) ? s->mw : s->mb;
    if (s->board[cell] != opp) return -1;
    if (in_shape(s->board, opp_marked, cell, opp) && has_normal(s->board, opp_marked, opp)) return -1;
    s->board[cell] = 0;
    s->pc_rem -= 1;
    if (count_pieces(s->board, opp) < LOSE_THRESHOLD) return 0;
    if (s->pc_rem > 0) return 0;
    s->pc_req = s->pc_rem = 0;
    s->player = -s->player;
    s->phase = PH_MOVEMENT;
    return 0;
}

/* rule_engine.py:417-457 */
static int do_counter(lzo_state* s, int cell) {
    if (s->phase != PH_COUNTER || cell < 0 || cell >= LZO_CELLS) return -1;
    int stuck = (int)-s->player;
    if (s->board[cell] != stuck) return -1;
    if (in_shape(s->board, NULL, cell, stuck) && has_normal(s->board, NULL, stuck)) return -1;
    s->board[cell] = 0;
    if (count_pieces(s->board, stuck) < LOSE_THRESHOLD) return 0;
    s->phase = PH_MOVEMENT;
    s->player = -s->player;
    return 0;
}

/* fast_apply_moves.cpp:595-753 apply_action (CPU) / fast_apply_moves_cuda.cu:548-744 (GPU).
 * Returns 0 if the rule function accepted the action, -1 if it was illegal.  In the GPU
 * semantics an illegal action leaves the state untouched but move_count is still bumped for
 * every kind except placement (placement bumps it inside the rule function), and an unknown
 * kind changes nothing but moves_since_capture. */
static int apply_code(const lzo_state* parent, const int32_t* code, lzo_state* out) {
    *out = *parent;
    int64_t phase_before = out->phase;
    int kind = code[0], primary = code[1], secondary = code[2];
    int rc = -1;
    switch (kind) {
        case K_PLACE:   rc = do_placement(out, primary); if (rc == 0) out->move_count += 1; break;
        case K_MARK:    rc = do_mark(out, primary); out->move_count += 1; break;
        case K_PROCESS: rc = do_process_removal(out); out->move_count += 1; break;
        case K_FORCED:  rc = do_forced(out, primary); out->move_count += 1; break;
        case K_MOVE:    rc = do_movement(out, primary, secondary); out->move_count += 1; break;
        case K_NOMOVES: rc = do_no_moves(out, primary); out->move_count += 1; break;
        case K_CAPTURE: rc = do_capture(out, primary); out->move_count += 1; break;
        case K_COUNTER: rc = do_counter(out, primary); out->move_count += 1; break;
        default: rc = -1; break;
    }
    /* src/move_generator.py:122-137 */
    if (phase_before == PH_PLACEMENT || phase_before == PH_MARK) {
        out->msc = 0;
    } else {
        int old_total = 0, new_total = 0;
        for (int i = 0; i < LZO_CELLS; ++i) { old_total += parent->board[i] != 0; new_total += out->board[i] != 0; }
        out->msc = (new_total < old_total) ? 0 : parent->msc + 1;
    }
    return rc;
}

int64_t lzo_apply_moves(const lzo_batch* in, int64_t B, const int32_t* codes,
                        const int64_t* parents, int64_t N, lzo_batch* out, int strict) {
    for (int64_t i = 0; i < N; ++i) {
        int64_t p = parents[i];
        if (p < 0 || p >= B) { if (strict) return -(i + 1); continue; }
        lzo_state ps, cs;
        load_state(in, p, &ps);
        int rc = apply_code(&ps, codes + i * 4, &cs);
        if (rc != 0 && strict) return -(i + 1);
        store_state(out, i, &cs);
    }
    return 0;
}

/* src/game_state.py:87-96,165-181 */
int lzo_game_status(const lzo_state* s) {
    if (s->phase == PH_MOVEMENT || s->phase == PH_CAPTURE || s->phase == PH_COUNTER) {
        if (count_pieces(s->board, 1) < LOSE_THRESHOLD) return -1;   /* white wins */
        if (count_pieces(s->board, -1) < LOSE_THRESHOLD) return 1;   /* black wins */
    }
    if (s->move_count >= MAX_MOVE_COUNT || s->msc >= NO_CAPTURE_LIMIT) return 2;
    return 0;
}

/* src/move_generator.py:24-70 + src/policy_batch.py:28-66 (220-d index), python semantics */
int lzo_legal_indices_py(const lzo_state* s, int* idx_out) {
    if (lzo_game_status(s) != 0) return 0;
    int n = 0;
    if (s->phase == PH_PLACEMENT) {
        for (int c = 0; c < LZO_CELLS; ++c) if (s->board[c] == 0) idx_out[n++] = c;
        return n;
    }
    if (s->phase == PH_REMOVAL) { idx_out[0] = 216; return 1; }
    int has_movement = 0;
    if (s->phase == PH_MOVEMENT) {
        for (int c = 0; c < LZO_CELLS; ++c) {
            if (s->board[c] != s->player) continue;
            int r = c / SIZE, cc = c % SIZE;
            for (int d = 0; d < 4; ++d) {
                int nr = r + DR[d], nc = cc + DC[d];
                if (nr < 0 || nr >= SIZE || nc < 0 || nc >= SIZE) continue;
                if (s->board[nr * SIZE + nc] == 0) { idx_out[n++] = 36 + c * 4 + d; has_movement = 1; }
            }
        }
        if (has_movement) return n;
    }
    int targets[LZO_CELLS], kind = 0;
    int m = selection_targets(s, has_movement, /*fallback_forced=*/0, targets, &kind);
    for (int i = 0; i < m; ++i) idx_out[n++] = 180 + targets[i];
    return n;
}

/* 220-d index -> action code for the current phase (v0/python/move_encoder.py:164-247) */
static int index_to_code(const lzo_state* s, int a, int32_t* code) {
    code[0] = K_INVALID; code[1] = code[2] = code[3] = -1;
    if (a < 0) return -1;
    if (a < 36) { code[0] = K_PLACE; code[1] = a; return 0; }
    if (a < 180) {
        int from = (a - 36) / 4, d = (a - 36) % 4;
        int nr = from / SIZE + DR[d], nc = from % SIZE + DC[d];
        code[0] = K_MOVE; code[1] = from; code[2] = d;
        code[3] = (nr >= 0 && nr < SIZE && nc >= 0 && nc < SIZE) ? nr * SIZE + nc : -1;
        return 0;
    }
    if (a < 216) {
        int cell = a - 180;
        code[1] = cell;
        switch (s->phase) {
            case PH_MARK: code[0] = K_MARK; break;
            case PH_CAPTURE: code[0] = K_CAPTURE; break;
            case PH_FORCED: code[0] = K_FORCED; break;
            case PH_COUNTER: code[0] = K_COUNTER; break;
            case PH_MOVEMENT: code[0] = K_NOMOVES; break;
            default: return -1;
        }
        return 0;
    }
    if (a == 216) { code[0] = K_PROCESS; return 0; }
    return -1;
}

int lzo_apply_index(const lzo_state* s, int action_index, lzo_state* out) {
    int32_t code[4];
    if (index_to_code(s, action_index, code) != 0) return -1;
    return apply_code(s, code, out);
}

/* v0/src/net/encoding.cpp:26-79 == src/neural_network.py:15-65 */
void lzo_states_to_model_input(const lzo_batch* in, int64_t B, float* out) {
    for (int64_t b = 0; b < B; ++b) {
        float* o = out + b * 11 * LZO_CELLS;
        const int8_t* board = in->board + b * LZO_CELLS;
        const uint8_t* mb = in->marks_black + b * LZO_CELLS;
        const uint8_t* mw = in->marks_white + b * LZO_CELLS;
        int64_t cur = in->current_player[b];
        int8_t cur8 = (int8_t)cur;           /* encoding.cpp:48: current cast to board dtype */
        int8_t neg8 = (int8_t)(-cur8);
        int is_black = (cur == 1);
        for (int i = 0; i < LZO_CELLS; ++i) {
            o[0 * LZO_CELLS + i] = (board[i] == cur8) ? 1.0f : 0.0f;
            o[1 * LZO_CELLS + i] = (board[i] == neg8) ? 1.0f : 0.0f;
            o[2 * LZO_CELLS + i] = (is_black ? mb[i] : mw[i]) ? 1.0f : 0.0f;
            o[3 * LZO_CELLS + i] = (is_black ? mw[i] : mb[i]) ? 1.0f : 0.0f;
        }
        for (int p = 1; p <= 7; ++p) {
            float v = (in->phase[b] == p) ? 1.0f : 0.0f;
            for (int i = 0; i < LZO_CELLS; ++i) o[(3 + p) * LZO_CELLS + i] = v;
        }
    }
}

/* v0/src/net/project_policy_logits_fast.cpp:16-164 (fp32) */
void lzo_project_policy(const float* lp1, const float* lp2, const float* lpmc,
                        const uint8_t* mask, int64_t B, int64_t pd, int64_t md, int64_t sd,
                        int64_t ad, float* probs, float* masked_logits) {
    const int64_t T = pd + md + sd + ad;
    const int bs = (int)llround(sqrt((double)pd));
    for (int64_t b = 0; b < B; ++b) {
        const float* p1 = lp1 + b * pd;
        const float* p2 = lp2 + b * pd;
        const float* pm = lpmc + b * pd;
        const uint8_t* mk = mask + b * T;
        float* pr = probs + b * T;
        float* ml = masked_logits + b * T;
        for (int64_t a = 0; a < T; ++a) {
            float v;
            if (a < pd) v = p1[a];
            else if (a < pd + md) {
                int from = (int)((a - pd) / 4), d = (int)((a - pd) % 4);
                int nr = from / bs + DR[d], nc = from % bs + DC[d];
                if (nr >= 0 && nr < bs && nc >= 0 && nc < bs) v = p2[from] + p1[nr * bs + nc];
                else v = -INFINITY;
            } else if (a < pd + md + sd) v = pm[a - pd - md];
            else v = 0.0f;
            ml[a] = mk[a] ? v : -INFINITY;
            pr[a] = 0.0f;
        }
        int has_legal = 0, has_finite = 0;
        float mx = -INFINITY;
        for (int64_t a = 0; a < T; ++a) {
            if (mk[a]) has_legal = 1;
            if (isfinite(ml[a])) { has_finite = 1; if (ml[a] > mx) mx = ml[a]; }
        }
        if (!has_legal) continue;
        if (has_finite) {
            /* softmax over the row (illegal = -inf => 0) */
            float sum = 0.0f;
            for (int64_t a = 0; a < T; ++a) { float e = (ml[a] == -INFINITY) ? 0.0f : expf(ml[a] - mx); pr[a] = e; sum += e; }
            for (int64_t a = 0; a < T; ++a) pr[a] = pr[a] / sum;
        } else {
            /* :153-160 -- legal entries of masked_logits become 0, probs stay 0 */
            for (int64_t a = 0; a < T; ++a) if (mk[a]) ml[a] = 0.0f;
        }
    }
}

/* v0/src/mcts/root_puct_fused.cu:44-116: sims serial bandit pulls, fp32, lowest index wins ties */
void lzo_root_puct(const float* priors, const float* leaf, const uint8_t* valid, int64_t R,
                   int64_t A, int64_t sims, float c, float* visits, float* value_sum,
                   float* root_values) {
    for (int64_t r = 0; r < R; ++r) {
        const float* p = priors + r * A;
        const float* lv = leaf + r * A;
        const uint8_t* vm = valid + r * A;
        float* vis = visits + r * A;
        float* vs = value_sum + r * A;
        for (int64_t a = 0; a < A; ++a) { vis[a] = 0.0f; vs[a] = 0.0f; }
        float total = 0.0f;
        for (int64_t s = 0; s < sims; ++s) {
            const float sqrt_total = sqrtf(total + 1.0f);
            float best = -INFINITY;
            int64_t best_idx = -1;
            for (int64_t a = 0; a < A; ++a) {
                if (!vm[a]) continue;
                const float v = vis[a];
                const float q = v > 0.0f ? (vs[a] / fmaxf(v, 1e-8f)) : 0.0f;
                const float u = c * p[a] * sqrt_total / (1.0f + v);
                const float score = q + u;
                if (score > best || (score == best && best_idx < 0)) { best = score; best_idx = a; }
            }
            if (best_idx >= 0) { vis[best_idx] += 1.0f; vs[best_idx] += lv[best_idx]; total += 1.0f; }
        }
        float sv = 0.0f, sw = 0.0f;
        for (int64_t a = 0; a < A; ++a) { sv += vis[a]; sw += vs[a]; }
        root_values[r] = sw / fmaxf(sv, 1.0f);
    }
}

/* ==========================================================================================
 * Variant-P tree search: v1/python/portable_mcts.py (== src/mcts.py with batch_K=1, vl=0)
 * ======================================================================================== */
typedef struct {
    lzo_state state;
    int parent;
    int first_child, n_children;
    int action_index;
    double prior;         /* float32 value widened (portable_mcts.py:471) */
    int visit_count;
    double value_sum;
    int player;
    int terminal, expanded, no_legal_terminal;
    double initial_value;
} lzo_node;

#define LZO_WAVE_MAX 64
struct lzo_tree {
    lzo_node* nodes;
    int n_nodes, cap;
    int root;
    double c;
    int path[1024];
    int path_len;
    int pending;   /* node index awaiting evaluation, -1 if none */
    int pending_is_root;
    int wave_eval[LZO_WAVE_MAX];   /* leaves of the current wave that await evaluation (src/mcts.py batch_K waves) */
    int wave_n;
};

static int tree_new_node(lzo_tree* t, const lzo_state* s, int parent, int action, double prior) {
    if (t->n_nodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 256;
        t->nodes = (lzo_node*)realloc(t->nodes, (size_t)t->cap * sizeof(lzo_node));
    }
    lzo_node* n = &t->nodes[t->n_nodes];
    memset(n, 0, sizeof(*n));
    n->state = *s;
    n->parent = parent;
    n->first_child = -1;
    n->action_index = action;
    n->prior = prior;
    n->player = (int)s->player;
    n->terminal = lzo_game_status(s) != 0;   /* portable_mcts.py:58-60 */
    return t->n_nodes++;
}

lzo_tree* lzo_tree_new(const lzo_state* root, double exploration_weight) {
    lzo_tree* t = (lzo_tree*)calloc(1, sizeof(lzo_tree));
    t->c = exploration_weight;
    t->pending = -1;
    t->root = tree_new_node(t, root, -1, -1, 1.0);
    return t;
}

void lzo_tree_free(lzo_tree* t) { if (t) { free(t->nodes); free(t); } }
int lzo_tree_node_count(const lzo_tree* t) { return t->n_nodes; }

/* portable_mcts.py:133-148 */
static double terminal_value(const lzo_state* s) {
    int st = lzo_game_status(s);
    if (st == 1 || st == -1) return (st == (int)s->player) ? 1.0 : -1.0;
    return 0.0;
}

/* portable_mcts.py:123-138 */
static void backup_path(lzo_tree* t, double leaf_value) {
    double value = leaf_value;
    for (int off = t->path_len - 1; off >= 0; --off) {
        lzo_node* n = &t->nodes[t->path[off]];
        n->visit_count += 1;
        n->value_sum += value;
        if (off > 0) {
            lzo_node* par = &t->nodes[t->path[off - 1]];
            if (par->player != n->player) value = -value;
        }
    }
}

int lzo_tree_prepare_root(lzo_tree* t) {
    lzo_node* r = &t->nodes[t->root];
    t->pending = -1;
    if (lzo_game_status(&r->state) != 0) { r->terminal = 1; return 0; }   /* :601-603 */
    if (r->expanded) return 0;
    t->pending = t->root;
    t->pending_is_root = 1;
    return 1;
}

/* portable_mcts.py:480-506 */
int lzo_tree_select(lzo_tree* t) {
    t->pending = -1;
    lzo_node* root = &t->nodes[t->root];
    if (root->terminal) return 0;
    int cur = t->root;
    t->path_len = 0;
    t->path[t->path_len++] = cur;
    for (;;) {
        lzo_node* n = &t->nodes[cur];
        if (!(n->expanded && n->n_children > 0 && !n->terminal)) break;
        double sqrt_total = sqrt((double)(n->visit_count > 1 ? n->visit_count : 1));
        double best = -INFINITY;
        int best_child = -1;
        for (int k = 0; k < n->n_children; ++k) {
            lzo_node* ch = &t->nodes[n->first_child + k];
            double q = 0.0;
            if (ch->visit_count > 0) {
                double mv = ch->value_sum / (double)ch->visit_count;
                q = (n->player == ch->player) ? mv : -mv;
            }
            double u = t->c * ch->prior * sqrt_total / (1.0 + (double)ch->visit_count);
            double score = q + u;
            if (score > best) { best = score; best_child = n->first_child + k; }
        }
        if (best_child < 0) break;
        cur = best_child;
        t->path[t->path_len++] = cur;
    }
    lzo_node* leaf = &t->nodes[cur];
    if (leaf->terminal) {                                                   /* :632-634 */
        backup_path(t, leaf->no_legal_terminal ? -1.0 : terminal_value(&leaf->state));
        return 0;
    }
    if (leaf->expanded && leaf->n_children == 0) {                           /* :635-638 */
        leaf->terminal = 1; leaf->no_legal_terminal = 1;
        backup_path(t, -1.0);
        return 0;
    }
    t->pending = cur;
    t->pending_is_root = 0;
    return 1;
}

void lzo_tree_pending_state(const lzo_tree* t, lzo_state* out) {
    *out = t->nodes[t->pending >= 0 ? t->pending : t->root].state;
}

/* portable_mcts.py:418-478 */
void lzo_tree_complete(lzo_tree* t, const float* priors220, float value, const float* noise,
                       float epsilon) {
    int ni = t->pending;
    if (ni < 0) return;
    t->pending = -1;
    lzo_state st = t->nodes[ni].state;
    int idx[80];
    int n = lzo_legal_indices_py(&st, idx);
    double ret;
    if (n == 0) {
        lzo_node* nd = &t->nodes[ni];
        nd->expanded = 1; nd->terminal = 1;
        nd->no_legal_terminal = lzo_game_status(&st) == 0;
        nd->initial_value = nd->no_legal_terminal ? -1.0 : terminal_value(&st);
        ret = nd->initial_value;
    } else {
        float pr[80];
        for (int k = 0; k < n; ++k) pr[k] = priors220[idx[k]];
        if (t->pending_is_root && noise != NULL && n > 1) {
            const float keep = (float)(1.0 - (double)epsilon);
            for (int k = 0; k < n; ++k) pr[k] = keep * pr[k] + epsilon * noise[k];
        }
        float sum = 0.0f;
        for (int k = 0; k < n; ++k) sum += pr[k];
        if (!isfinite(sum) || sum <= 0.0f) { for (int k = 0; k < n; ++k) pr[k] = 1.0f / (float)n; }
        else { for (int k = 0; k < n; ++k) pr[k] = pr[k] / sum; }
        int first = t->n_nodes;
        for (int k = 0; k < n; ++k) {
            lzo_state cs;
            lzo_apply_index(&st, idx[k], &cs);
            tree_new_node(t, &cs, ni, idx[k], (double)pr[k]);
        }
        lzo_node* nd = &t->nodes[ni];   /* re-fetch: arena may have moved */
        nd->first_child = first; nd->n_children = n; nd->expanded = 1;
        nd->initial_value = (double)value;
        ret = nd->initial_value;
    }
    if (!t->pending_is_root) backup_path(t, ret);   /* roots are expanded without backup (:604-616) */
}

/* portable_mcts.py:302-317 */
void lzo_tree_root_noise(lzo_tree* t, const float* noise, float epsilon) {
    lzo_node* r = &t->nodes[t->root];
    if (!r->expanded || r->n_children <= 1) return;
    float pr[80];
    const float keep = (float)(1.0 - (double)epsilon);
    float sum = 0.0f;
    for (int k = 0; k < r->n_children; ++k) {
        pr[k] = keep * (float)t->nodes[r->first_child + k].prior + epsilon * noise[k];
        sum += pr[k];
    }
    float denom = sum < 1e-8f ? 1e-8f : sum;
    for (int k = 0; k < r->n_children; ++k) t->nodes[r->first_child + k].prior = (double)(pr[k] / denom);
}

/* ---- wave-batched selection of the legacy search (src/mcts.py:318-497, batch_K leaves per wave, no virtual loss) ----
 * Literal restatement: every attempt walks down from the root choosing the best child that is not banned for this
 * attempt; a leaf already reserved in this wave is banned at its parent and the walk moves to the next best sibling
 * or, when a node has no candidate left, backs up one level and bans that node (:341-420).  Backups go through the
 * parent links like MCTSNode.backpropagate (:106-131). */
static void backup_from(lzo_tree* t, int ni, double leaf_value) {
    double value = leaf_value;
    while (ni >= 0) {
        lzo_node* n = &t->nodes[ni];
        n->visit_count += 1;
        n->value_sum += value;
        const int par = (ni == t->root) ? -1 : n->parent;
        if (par >= 0 && t->nodes[par].player != n->player) value = -value;
        ni = par;
    }
}

static int in_list(const int* a, int n, int v) {
    for (int i = 0; i < n; ++i) if (a[i] == v) return 1;
    return 0;
}

/* MCTSNode.get_best_child_excluding (:205-222): first maximum in child order among the children not banned */
static int best_child_excluding(const lzo_tree* t, int ni, const int* banned, int nb) {
    const lzo_node* n = &t->nodes[ni];
    if (!n->expanded || n->n_children <= 0) return -1;
    const double sqrt_total = sqrt((double)(n->visit_count > 1 ? n->visit_count : 1));
    double best = -INFINITY;
    int best_child = -1;
    for (int k = 0; k < n->n_children; ++k) {
        const int ci = n->first_child + k;
        if (in_list(banned, nb, ci)) continue;
        const lzo_node* ch = &t->nodes[ci];
        double q = 0.0;
        if (ch->visit_count > 0) {
            const double mv = ch->value_sum / (double)ch->visit_count;
            q = (n->player == ch->player) ? mv : -mv;
        }
        const double u = t->c * ch->prior * sqrt_total / (1.0 + (double)ch->visit_count);
        const double score = q + u;
        if (score > best) { best = score; best_child = ci; }
    }
    return best_child;
}

/* Collect up to `to_collect` distinct leaves (:333-425), then the terminal fast paths in leaf order (:432-459).
 * Returns the number of simulations consumed (= leaves collected); the leaves that need the network are left in
 * wave_eval[0..wave_n). */
/* MAX_BACKTRACK_STEPS of src/mcts.py:337 (128); adjustable so that a test can make the limit bite on small trees */
static int g_max_backtrack = 128;
void lzo_set_max_backtrack(int n) { g_max_backtrack = n > 0 ? n : 128; }

int lzo_tree_select_wave(lzo_tree* t, int to_collect) {
    t->wave_n = 0;
    if (to_collect > LZO_WAVE_MAX) to_collect = LZO_WAVE_MAX;
    const lzo_node* rootn = &t->nodes[t->root];
    if (rootn->terminal || !rootn->expanded || rootn->n_children == 0) return 0;
    int leaves[LZO_WAVE_MAX];
    int collected = 0, attempts = 0;
    const int max_attempts = (8 * to_collect > 64) ? 8 * to_collect : 64;
    int* banned = (int*)malloc(sizeof(int) * 4096);
    while (collected < to_collect && attempts < max_attempts) {
        ++attempts;
        int node = t->root, nb = 0, backtrack = 0;
        for (;;) {
            const lzo_node* n = &t->nodes[node];
            if (!n->expanded || n->terminal) {                       /* candidate leaf */
                if (in_list(leaves, collected, node)) {              /* reserved in this wave */
                    if (node == t->root) break;
                    const int parent = n->parent;
                    if (nb < 4096) banned[nb++] = node;
                    const int alt = best_child_excluding(t, parent, banned, nb);
                    if (alt < 0) { node = parent; if (++backtrack > g_max_backtrack) break; continue; }
                    node = alt;
                    continue;
                }
                leaves[collected++] = node;
                break;
            }
            const int child = best_child_excluding(t, node, banned, nb);
            if (child < 0) {
                if (node == t->root) break;
                if (nb < 4096) banned[nb++] = node;
                node = n->parent;
                if (++backtrack > g_max_backtrack) break;
                continue;
            }
            node = child;
        }
    }
    free(banned);
    for (int i = 0; i < collected; ++i) {
        lzo_node* leaf = &t->nodes[leaves[i]];
        if (leaf->terminal) {                                        /* known terminal, or game over on arrival */
            backup_from(t, leaves[i], leaf->no_legal_terminal ? -1.0 : terminal_value(&leaf->state));
            continue;
        }
        int idx[80];
        if (lzo_legal_indices_py(&leaf->state, idx) == 0) {          /* no legal move: the mover loses */
            leaf->terminal = 1; leaf->no_legal_terminal = 1; leaf->expanded = 1;
            backup_from(t, leaves[i], -1.0);
            continue;
        }
        t->wave_eval[t->wave_n++] = leaves[i];
    }
    return collected;
}

int lzo_tree_wave_count(const lzo_tree* t) { return t->wave_n; }
void lzo_tree_wave_state(const lzo_tree* t, int j, lzo_state* out) { *out = t->nodes[t->wave_eval[j]].state; }

/* :478-497: expand every evaluated leaf with its priors and back its value up, in leaf order */
void lzo_tree_complete_wave(lzo_tree* t, const float* priors220, const float* values) {
    for (int j = 0; j < t->wave_n; ++j) {
        const int ni = t->wave_eval[j];
        lzo_state st = t->nodes[ni].state;
        int idx[80];
        const int n = lzo_legal_indices_py(&st, idx);
        float pr[80];
        float sum = 0.0f;
        for (int k = 0; k < n; ++k) { pr[k] = priors220[(size_t)j * 220 + idx[k]]; sum += pr[k]; }
        if (!isfinite(sum) || sum <= 0.0f) { for (int k = 0; k < n; ++k) pr[k] = 1.0f / (float)n; }
        else { for (int k = 0; k < n; ++k) pr[k] = pr[k] / sum; }
        const int first = t->n_nodes;
        for (int k = 0; k < n; ++k) {
            lzo_state cs;
            lzo_apply_index(&st, idx[k], &cs);
            tree_new_node(t, &cs, ni, idx[k], (double)pr[k]);
        }
        lzo_node* nd = &t->nodes[ni];
        nd->first_child = first; nd->n_children = n; nd->expanded = 1;
        nd->initial_value = (double)values[j];
        backup_from(t, ni, (double)values[j]);
    }
    t->wave_n = 0;
}

int lzo_tree_root_terminal(const lzo_tree* t) {
    const lzo_node* r = &t->nodes[t->root];
    return r->terminal || r->n_children == 0;
}

int lzo_tree_root_children(const lzo_tree* t, int* action_idx, int* visits, double* value_sum,
                           float* prior, int* child_player) {
    const lzo_node* r = &t->nodes[t->root];
    for (int k = 0; k < r->n_children; ++k) {
        const lzo_node* ch = &t->nodes[r->first_child + k];
        if (action_idx) action_idx[k] = ch->action_index;
        if (visits) visits[k] = ch->visit_count;
        if (value_sum) value_sum[k] = ch->value_sum;
        if (prior) prior[k] = (float)ch->prior;
        if (child_player) child_player[k] = ch->player;
    }
    return r->n_children;
}

int lzo_tree_root_visits(const lzo_tree* t) { return t->nodes[t->root].visit_count; }
double lzo_tree_root_value_sum(const lzo_tree* t) { return t->nodes[t->root].value_sum; }
int lzo_tree_root_player(const lzo_tree* t) { return t->nodes[t->root].player; }

int lzo_tree_advance(lzo_tree* t, int action_index) {
    lzo_node* r = &t->nodes[t->root];
    for (int k = 0; k < r->n_children; ++k) {
        if (t->nodes[r->first_child + k].action_index == action_index) {
            t->root = r->first_child + k;
            t->nodes[t->root].parent = -1;
            t->pending = -1;
            return 1;
        }
    }
    return 0;
}
