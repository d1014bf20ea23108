/*
 * lz_oracle.h -- CPU ORACLE for the Liuzhou self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's algorithm
 * (kuailehaha/liuzhou) used as the parity checker.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (liuzhou_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * Pinned against (see tests/test_oracle_golden.py, oracle/gen_golden.py):
 *   - src/rule_engine.py + src/move_generator.py random playouts (reachable states, all 7 phases)
 *   - v1/python/portable_mcts.py visit counts (recorded evaluator outputs)
 *   - the reference's own v0_core CPU ops built under oracle/_ref (synthetic "garbage" states)
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef LZ_ORACLE_H
#define LZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZO_CELLS 36
#define LZO_META 4

/* Struct-of-arrays view of a state batch: exactly the 12 tensors of
 * v0/include/v0/tensor_state_batch.hpp:11-35 / v1/python/mcts_gpu.py:40-57. */
typedef struct {
    int8_t*  board;        /* [B,36] in {-1,0,1} */
    uint8_t* marks_black;  /* [B,36] bool */
    uint8_t* marks_white;  /* [B,36] bool */
    int64_t* phase;        /* [B] 1..7 */
    int64_t* current_player; /* [B] +1 black / -1 white */
    int64_t* pending_marks_required;
    int64_t* pending_marks_remaining;
    int64_t* pending_captures_required;
    int64_t* pending_captures_remaining;
    int64_t* forced_removals_done;
    int64_t* move_count;
    int64_t* moves_since_capture;
} lzo_batch;

/* One scalar state (src/game_state.py:27-65). */
typedef struct {
    int8_t  board[LZO_CELLS];
    uint8_t mb[LZO_CELLS];
    uint8_t mw[LZO_CELLS];
    int64_t phase, player;
    int64_t pm_req, pm_rem, pc_req, pc_rem;
    int64_t forced, move_count, msc;
} lzo_state;

/* ---- rule ops (tensor semantics; v0/src/game/fast_legal_mask.cpp:253-418) ---- */
void lzo_encode_actions(const lzo_batch* in, int64_t B,
                        int64_t placement_dim, int64_t movement_dim,
                        int64_t selection_dim, int64_t auxiliary_dim,
                        uint8_t* mask /*[B,T]*/, int32_t* meta /*[B,T,4]*/);

/* v0/src/game/fast_apply_moves.cpp:595-753 (strict=1: CPU semantics, returns -(i+1) at the
 * first illegal action) / fast_apply_moves_cuda.cu:548-744 (strict=0: illegal => no-op). */
int64_t lzo_apply_moves(const lzo_batch* in, int64_t B,
                        const int32_t* codes /*[N,4]*/, const int64_t* parents /*[N]*/,
                        int64_t N, lzo_batch* out, int strict);

/* src/game_state.py:87-96,165-181: 0 = running, 1 = black wins, -1 = white wins, 2 = draw. */
int lzo_game_status(const lzo_state* s);

/* src/move_generator.py:24-70 (python semantics: [] when game over, no forced-removal
 * fallback).  Returns count; writes ascending 220-d indices. */
int lzo_legal_indices_py(const lzo_state* s, int* idx_out /*[<=72]*/);
/* src/move_generator.py:73-139 by 220-d index.  Returns 0 ok / -1 illegal. */
int lzo_apply_index(const lzo_state* s, int action_index, lzo_state* out);

/* v0/src/net/encoding.cpp:26-79 / src/neural_network.py:15-65 */
void lzo_states_to_model_input(const lzo_batch* in, int64_t B, float* out /*[B,11,36]*/);

/* v0/src/net/project_policy_logits_fast.cpp:16-164 */
void lzo_project_policy(const float* lp1, const float* lp2, const float* lpmc,
                        const uint8_t* mask, int64_t B,
                        int64_t placement_dim, int64_t movement_dim,
                        int64_t selection_dim, int64_t auxiliary_dim,
                        float* probs, float* masked_logits);

/* v0/src/mcts/root_puct_fused.cu:12-117 (== v0/src/bindings/module.cpp:180-245) */
void lzo_root_puct(const float* priors, const float* leaf, const uint8_t* valid,
                   int64_t R, int64_t A, int64_t sims, float c,
                   float* visits, float* value_sum, float* root_values);

/* ---- tree search, variant P (v1/python/portable_mcts.py:264-746; == src/mcts.py batch_K=1) ---- */
typedef struct lzo_tree lzo_tree;
lzo_tree* lzo_tree_new(const lzo_state* root, double exploration_weight);
void lzo_tree_free(lzo_tree* t);
/* 1 = root needs an evaluation (pending state = root), 0 = no (terminal or already expanded) */
int  lzo_tree_prepare_root(lzo_tree* t);
/* descend once; 1 = a leaf is pending evaluation, 0 = terminal leaf was backed up / root terminal */
int  lzo_tree_select(lzo_tree* t);
void lzo_tree_pending_state(const lzo_tree* t, lzo_state* out);
/* complete the pending node with the network's 220-d priors + value.  For a root completion,
 * noise (length = #legal, in ascending action order) may be given with epsilon. */
void lzo_tree_complete(lzo_tree* t, const float* priors220, float value,
                       const float* noise, float epsilon);
/* re-noise an already expanded (reused) root: portable_mcts.py:302-317 */
void lzo_tree_root_noise(lzo_tree* t, const float* noise, float epsilon);
int  lzo_tree_root_terminal(const lzo_tree* t);
int  lzo_tree_root_children(const lzo_tree* t, int* action_idx, int* visits,
                            double* value_sum, float* prior, int* child_player);
int  lzo_tree_root_visits(const lzo_tree* t);
double lzo_tree_root_value_sum(const lzo_tree* t);
int  lzo_tree_root_player(const lzo_tree* t);
/* src/mcts.py:577-592 / portable_mcts.py:74-87: 1 if the child subtree was kept */
int  lzo_tree_advance(lzo_tree* t, int action_index);
int  lzo_tree_node_count(const lzo_tree* t);

#ifdef __cplusplus
}
#endif
#endif
