/* liuzhou_scalar.h -- the SCALAR rule surface of the reference's `v0_core` module, as a C ABI.
 *
 * The reference binds, next to its batched tensor operators, one-state-at-a-time rule functions over a `GameState`
 * class (v0/src/bindings/module.cpp:974-1110: generate_* / apply_* / process_phase2_removals / handle_no_moves_phase3 /
 * generate_all_legal_moves_struct / apply_move_struct ...; bodies v0/src/rules/rule_engine.cpp:194-725,
 * v0/src/moves/move_generator.cpp:149-439, v0/src/game/game_state.cpp:24-79).  `v0/python/move_generator.py` and the
 * rule tests / tools use them on the host, one state per call.  They are host functions in the reference and they are
 * host functions here: exported by libliuzhou_host.so only (csrc/lz_scalar.cpp, g++), on the same 36-bit bitboard rules the
 * gfx950 kernels include (csrc/lz_rules.h).  Not part of the device library; nothing on the self-play path calls them.
 *
 * Reference-side binding a maintainer would add: the PyBind11 lambdas of module.cpp:1008-1110 calling these instead of
 * v0::Generate... / v0::Apply...; ours is liuzhou_amd/v0_scalar.py (ctypes).
 *
 * Conventions: plain structs and pointers, no allocation, no global state, thread-safe.  Return value LZ_OK (0) or a
 * negative LzStatus of liuzhou_hip.h; LZ_ERR_ILLEGAL (-5) = the reference function throws std::runtime_error for this
 * input, and *reason (if not NULL) receives which check failed (LzScalarReason) so that the caller can word the error.
 * Cells are r * 6 + c; a cell outside 0..35 stands for an out-of-board coordinate. */
#ifndef LIUZHOU_SCALAR_H_
#define LIUZHOU_SCALAR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef LZ_API
#define LZ_API __attribute__((visibility("default")))
#endif

/* v0::GameState (v0/include/v0/game_state.hpp:97-147), flattened. */
typedef struct LzScalarState {
    int8_t  board[36];              /* -1 white, 0 empty, +1 black */
    uint8_t marks_black[36];        /* 0 / 1 */
    uint8_t marks_white[36];
    int32_t phase;                  /* 1..7 (v0::Phase) */
    int32_t current_player;         /* +1 black, -1 white */
    int32_t forced_removals_done;
    int32_t move_count;
    int32_t pending_marks_required, pending_marks_remaining;
    int32_t pending_captures_required, pending_captures_remaining;
    int32_t moves_since_capture;    /* carried by the reference's struct, not exposed to Python (module.cpp:974-1006) */
} LzScalarState;

/* v0::MoveRecord (v0/include/v0/move_generator.hpp:25-47): phase, v0::ActionType 1..8, primary / secondary cell (-1). */
typedef struct LzScalarMove {
    int32_t phase;
    int32_t action_type;
    int32_t primary;
    int32_t secondary;
} LzScalarMove;

/* What lz_scalar_generate lists (ascending cell order = the reference's row-major scans; movement: per piece up, down,
 * left, right).  Output units: cells (1 int32) -- movement: (from, to) pairs (2 int32) -- moves: LzScalarMove (4 int32). */
typedef enum LzScalarList {
    LZ_LIST_PLACEMENT_POSITIONS = 0,   /* GeneratePlacementPositions        rule_engine.cpp:210-224 */
    LZ_LIST_MARK_TARGETS = 1,          /* GenerateMarkTargets               rule_engine.cpp:281-308 */
    LZ_LIST_MOVEMENT_MOVES = 2,        /* GenerateMovementMoves             rule_engine.cpp:397-419 */
    LZ_LIST_CAPTURE_TARGETS = 3,       /* GenerateCaptureTargets            rule_engine.cpp:481-499 */
    LZ_LIST_FORCED_REMOVAL_MOVES = 4,  /* GenerateForcedRemovalMoves        move_generator.cpp:149-175 (moves) */
    LZ_LIST_NO_MOVES_OPTIONS = 5,      /* GenerateNoMovesOptions            move_generator.cpp:177-207 (moves) */
    LZ_LIST_COUNTER_REMOVAL_MOVES = 6, /* GenerateCounterRemovalMoves       move_generator.cpp:209-240 (moves) */
    LZ_LIST_ALL_LEGAL_MOVES = 7        /* GenerateAllLegalMoves             move_generator.cpp:242-297 (moves) */
} LzScalarList;

/* Transitions of lz_scalar_apply; a / b are cells (b only for the movement step).  None of them touches move_count /
 * moves_since_capture -- lz_scalar_apply_move does (= ApplyMove: phase / action-type match, then the bookkeeping of
 * move_generator.cpp:418-431). */
typedef enum LzScalarStep {
    LZ_STEP_PLACEMENT = 1,         /* ApplyPlacementMove(state, a)            rule_engine.cpp:226-279 */
    LZ_STEP_MARK = 2,              /* ApplyMarkSelection(state, a)            rule_engine.cpp:310-358 */
    LZ_STEP_PROCESS_REMOVAL = 3,   /* ProcessPhase2Removals(state)            rule_engine.cpp:360-395 */
    LZ_STEP_MOVEMENT = 4,          /* ApplyMovementMove(state, (a, b))        rule_engine.cpp:429-479 */
    LZ_STEP_CAPTURE = 5,           /* ApplyCaptureSelection(state, a)         rule_engine.cpp:501-547 */
    LZ_STEP_FORCED_REMOVAL = 6,    /* ApplyForcedRemoval(state, a)            rule_engine.cpp:549-595 */
    LZ_STEP_NO_MOVES = 7,          /* HandleNoMovesPhase3(state, a)           rule_engine.cpp:597-637 */
    LZ_STEP_COUNTER_REMOVAL = 8    /* ApplyCounterRemovalPhase3(state, a)     rule_engine.cpp:639-680 */
} LzScalarStep;

typedef enum LzScalarReason {
    LZ_WHY_NONE = 0,
    LZ_WHY_PHASE = 1,              /* the state is not in the phase the function serves */
    LZ_WHY_OUT_OF_BOARD = 2,
    LZ_WHY_OCCUPIED = 3,           /* placement on a piece / movement onto a piece */
    LZ_WHY_MARKED_BY_OPPONENT = 4, /* placement on a cell the opponent marked */
    LZ_WHY_NOTHING_PENDING = 5,    /* no mark / capture left to make */
    LZ_WHY_NOT_OPPONENT_PIECE = 6, /* the target is not a piece of the side that must lose one */
    LZ_WHY_ALREADY_MARKED = 7,
    LZ_WHY_IN_SHAPE = 8,           /* the target is part of a square / line while ordinary pieces remain (or, forced removal: at all) */
    LZ_WHY_NOT_OWN_PIECE = 9,      /* movement from a cell that is not the mover's piece */
    LZ_WHY_NOT_ONE_STEP = 10,      /* movement that is not one orthogonal step */
    LZ_WHY_FORCED_ORDER = 11,      /* forced removal by the wrong side / after both were made */
    LZ_WHY_MOVE_PHASE_MISMATCH = 12, /* ApplyMove: move.phase != state.phase */
    LZ_WHY_MOVE_TYPE = 13          /* ApplyMove: the action type is not one the phase allows */
} LzScalarReason;

/* List `what` for `state` into out[0 .. cap) (int32 units as above); *count = units written.  144 movement pairs are the
 * largest list (288 int32).  LZ_ERR_ARG: NULL pointers, unknown list, cap too small. */
LZ_API int lz_scalar_generate(const LzScalarState* state, int what, int32_t* out, int32_t cap, int32_t* count);

/* HasLegalMovementMoves (rule_engine.cpp:421-427): LZ_ERR_ILLEGAL (LZ_WHY_PHASE) outside the movement phase. */
LZ_API int lz_scalar_has_movement(const LzScalarState* state, int32_t* has_moves, int32_t* reason);

/* One transition; `next` may alias `state`.  On LZ_ERR_ILLEGAL `next` is left untouched. */
LZ_API int lz_scalar_apply(const LzScalarState* state, int step, int32_t a, int32_t b, LzScalarState* next, int32_t* reason);

/* ApplyMove(state, move) (move_generator.cpp:360-432): the transition the record names + move_count / moves_since_capture. */
LZ_API int lz_scalar_apply_move(const LzScalarState* state, const LzScalarMove* move, LzScalarState* next, int32_t* reason);

/* GetWinner / IsGameOver (game_state.cpp:58-79): *winner = +1 / -1 / 0 (none), *game_over = 0 / 1. */
LZ_API int lz_scalar_status(const LzScalarState* state, int32_t* winner, int32_t* game_over);

/* IsPieceInShape(state, r, c, player_value, marks of that player taken from the state or ignored)
 * (rule_engine.cpp:194-208): *in_shape = 0 / 1; an out-of-board cell or a cell without that player's piece gives 0. */
LZ_API int lz_scalar_piece_in_shape(const LzScalarState* state, int32_t cell, int32_t player, int32_t use_marks, int32_t* in_shape);

#ifdef __cplusplus
}
#endif
#endif  /* LIUZHOU_SCALAR_H_ */
