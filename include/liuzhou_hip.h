/*
 * liuzhou_hip.h -- C ABI of libliuzhou_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the self-play hot path of kuailehaha/liuzhou.  Every entry point
 * replaces one operator of the reference's PyBind11 module `v0_core`
 * (v0/src/bindings/module.cpp:874-1482) or one stage of the v1 wave loop
 * (v1/python/self_play_gpu_runner.py:159-256, v1/python/mcts_gpu.py:1249-1457); the replaced
 * reference interface is cited at each declaration (paths relative to the reference repo).
 *
 * Conventions (same for all entry points):
 *   - plain pointers + sizes, all pointers are DEVICE pointers unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - no allocation, no host synchronisation inside -> every call is hipGraph-capturable;
 *   - returns LZ_OK (0) or a negative LzStatus; on error nothing has been launched;
 *   - inputs are borrowed, outputs are caller-allocated with the documented shapes;
 *   - tensors are dense row-major; state batches use the reference's 12-tensor SoA layout
 *     (v0/include/v0/tensor_state_batch.hpp:11-35, v1/python/mcts_gpu.py:40-57).
 */
#ifndef LIUZHOU_HIP_H
#define LIUZHOU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define LZ_API
#else
#define LZ_API __attribute__((visibility("default")))
#endif

typedef enum LzStatus {
    LZ_OK = 0,
    LZ_ERR_ARG = -1,          /* null pointer / negative size / inconsistent dims */
    LZ_ERR_UNSUPPORTED = -2,  /* dims outside what the kernels are built for */
    LZ_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after launch */
    LZ_ERR_ALIGN = -4,        /* pointer not aligned as documented */
    LZ_ERR_ILLEGAL = -5       /* host library only: an action is illegal for its state (the reference's CPU path raises
                                 there, fast_apply_moves.cpp:264-470; the device path is a silent no-op) */
} LzStatus;

/* Two builds export this ABI:
 *   libliuzhou_hip.so  (hipcc, gfx950): everything below, pointers are DEVICE memory, launches on `stream`;
 *   libliuzhou_host.so (g++, csrc/lz_host.cpp): the v0_core OPERATOR subset (lz_encode_actions_fast .. lz_finalize_
 *     trajectory_inplace incl. lz_root_pack_*), pointers are HOST memory, `stream` is ignored -- what the Python
 *     `v0_core` module dispatches CPU tensors to, as the reference extension does (fast_legal_mask.cpp:453).
 * The search engines, the network kernels, the RNG and the trainer / codec kernels exist in the device build only. */

/* 12-tensor state batch.  board int8[B,36] in {-1,0,1}; marks bool(uint8)[B,36]; the rest int64[B].
 * board / marks rows must be 4-byte aligned (they are for any contiguous torch tensor). */
typedef struct LzStateSoA {
    int8_t*  board;
    uint8_t* marks_black;
    uint8_t* marks_white;
    int64_t* phase;
    int64_t* current_player;
    int64_t* pending_marks_required;
    int64_t* pending_marks_remaining;
    int64_t* pending_captures_required;
    int64_t* pending_captures_remaining;
    int64_t* forced_removals_done;
    int64_t* move_count;
    int64_t* moves_since_capture;
} LzStateSoA;

LZ_API const char* lz_version(void);
LZ_API const char* lz_status_string(int status);

/* ---- rule operators ------------------------------------------------------------------------- */

/* v0_core.encode_actions_fast  (module.cpp:1294-1310; fast_legal_mask.cpp:253-418, _cuda.cu:282-485)
 * mask uint8[B,T], metadata int32[B,T,4] with T = 36+144+36+auxiliary_dim; both fully written
 * (illegal entries: mask 0, metadata -1).  placement/movement/selection dims must be 36/144/36. */
LZ_API int lz_encode_actions_fast(const LzStateSoA* states, int64_t batch,
                                  int64_t placement_dim, int64_t movement_dim,
                                  int64_t selection_dim, int64_t auxiliary_dim,
                                  uint8_t* mask, int32_t* metadata, void* stream);

/* v0_core.batch_apply_moves  (module.cpp:1311-1327; fast_apply_moves_cuda.cu:548-744)
 * child i = apply(action_codes[i], states[parent_indices[i]]).  Illegal action => child == parent
 * with the reference GPU bookkeeping (move_count bumped for every kind except placement).
 * A parent index outside [0,batch) leaves child i untouched. */
LZ_API int lz_batch_apply_moves(const LzStateSoA* states, int64_t batch,
                                const int32_t* action_codes /*[N,4]*/,
                                const int64_t* parent_indices /*[N]*/, int64_t num_actions,
                                const LzStateSoA* out /*[N]*/, void* stream);

/* in-place variant (fast_apply_moves_cuda.cu:746-917): slot_indices must be unique. */
LZ_API int lz_batch_apply_moves_inplace(const LzStateSoA* states, int64_t batch,
                                        const int32_t* action_codes /*[N,4]*/,
                                        const int64_t* slot_indices /*[N]*/, int64_t num_actions,
                                        void* stream);

/* v0_core.states_to_model_input  (module.cpp:1286-1293; v0/src/net/encoding.cpp:26-79)
 * out float32[B,11,6,6]: own, opp, own marks, opp marks, 7 phase one-hot planes. */
LZ_API int lz_states_to_model_input(const int8_t* board, const uint8_t* marks_black,
                                    const uint8_t* marks_white, const int64_t* phase,
                                    const int64_t* current_player, int64_t batch,
                                    float* out, void* stream);

/* v0_core.project_policy_logits_fast  (module.cpp:1328-1338; project_policy_logits_fast.cpp:16-164)
 * float32 heads [B,36] x3 + legal mask uint8[B,T] -> probs, masked_logits float32[B,T]. */
LZ_API int lz_project_policy_logits_fast(const float* log_p1, const float* log_p2,
                                         const float* log_pmc, const uint8_t* legal_mask,
                                         int64_t batch, int64_t placement_dim, int64_t movement_dim,
                                         int64_t selection_dim, int64_t auxiliary_dim,
                                         float* probs, float* masked_logits, void* stream);

/* ---- root search operators (variant R: v1/python/mcts_gpu.py:1249-1457) -------------------- */

/* Row compaction at the heart of v0_core.root_pack_sparse_actions (module.cpp:247-363): for every
 * row b the legal action indices are packed to the left (ascending), with their priors renormalised
 * over the row (sum clamped at 1e-8) and their action codes.  Fixed capacity `cap` (>= max legal
 * count, 80 is always enough) keeps it sync-free; the Python operator slices to the reference's
 * data-dependent [R, Amax] shapes.
 *   counts int32[B]; legal_index int32[B,cap] (-1 pad); priors float32[B,cap] (0 pad);
 *   codes int32[B,cap,4] (0 pad). */
LZ_API int lz_root_pack_rows(const uint8_t* legal_mask, const float* probs, const int32_t* metadata,
                             int64_t batch, int64_t total_dim, int64_t cap,
                             int32_t* counts, int32_t* legal_index, float* priors, int32_t* codes,
                             void* stream);

/* v0_core.root_pack_sparse_actions (module.cpp:247-363, :1357-1362) through the C ABI.  Its output shapes [R, Amax]
 * depend on the data, so it is a two-call protocol with ONE host read in between (the reference reads R and Amax back
 * too, module.cpp:295,310):
 *   1. lz_root_pack_rows (above) -> fixed-capacity rows, then
 *      lz_root_pack_plan(counts, B, rank int32[B], child_off int64[B], sizes int64[3]) -- all device memory:
 *        rank[b] = index of row b among the non-terminal rows (-1: no legal action), child_off[b] = number of legal
 *        actions in the rows before b, sizes = {R = non-terminal rows, Amax = largest count, N = sum of counts};
 *   2. the caller reads `sizes`, allocates the ten outputs and calls
 *      lz_root_pack_fill(...): terminal_mask uint8[B], valid_root_indices int64[R], counts int64[R],
 *        valid_mask uint8[R,Amax], legal_index_mat int64[R,Amax] (0 pad), priors_mat float32[R,Amax] (0 pad),
 *        action_code_mat int32[R,Amax,4] (0 pad), pack_flat_idx int64[N] (row-major positions of the valid entries),
 *        action_codes_all int32[N,4], parent_indices_all int64[N] (row b of every entry) -- the reference's 10-tuple.
 * Pointers of empty outputs (R = 0 / N = 0) may be NULL.  The int32x4 code arrays must be 16-byte aligned. */
LZ_API int lz_root_pack_plan(const int32_t* counts, int64_t batch, int32_t* rank, int64_t* child_off, int64_t* sizes,
                             void* stream);
LZ_API int lz_root_pack_fill(const int32_t* counts, const int32_t* legal_index, const float* priors,
                             const int32_t* codes, const int32_t* rank, const int64_t* child_off, int64_t batch,
                             int64_t cap, int64_t R, int64_t Amax, int64_t N, uint8_t* terminal_mask,
                             int64_t* valid_root_indices, int64_t* counts_out, uint8_t* valid_mask,
                             int64_t* legal_index_mat, float* priors_mat, int32_t* action_code_mat,
                             int64_t* pack_flat_idx, int32_t* action_codes_all, int64_t* parent_indices_all,
                             void* stream);

/* v0_core.root_puct_allocate_visits  (module.cpp:1349-1356; root_puct_fused.cu:12-117)
 * fp32 bandit: `num_simulations` serial pulls per root, lowest index wins ties.  A <= 256.
 * The one entry point of the HIP build that owns device memory: the first eager call per (device, stream) allocates
 * 16 bytes per root of scratch for the width-binned kernel (rows of <= 8 / <= 16 / <= 32 valid actions share a wave eight /
 * four / two at a time) and keeps it; a call on a capturing stream uses what exists (an eager call with the same `num_roots` first),
 * otherwise the neighbours-pair-up kernel.  Outputs do not depend on which kernel ran (bit-identical). */
LZ_API int lz_root_puct_allocate_visits(const float* priors, const float* leaf_values,
                                        const uint8_t* valid_mask, int64_t num_roots,
                                        int64_t num_actions, int64_t num_simulations,
                                        float exploration_weight, float* visits, float* value_sum,
                                        float* root_values, void* stream);
/* The same with caller-owned scratch (`workspace` >= *bytes of lz_root_puct_workspace_bytes(num_roots, &bytes) of device
 * memory, 4-byte aligned): allocation-free like every other entry point, so it can be captured on any stream -- what the fused
 * root search of liuzhou_amd/root_search_fused.py launches. */
LZ_API int lz_root_puct_workspace_bytes(int64_t num_roots, int64_t* bytes);
LZ_API int lz_root_puct_allocate_visits_ws(const float* priors, const float* leaf_values,
                                           const uint8_t* valid_mask, int64_t num_roots, int64_t num_actions,
                                           int64_t num_simulations, float exploration_weight, float* visits,
                                           float* value_sum, float* root_values, void* workspace,
                                           int64_t workspace_bytes, void* stream);

/* v0_core.root_finalize_from_visits  (module.cpp:1374-1386, :441-535) fused with the v1 sampling
 * step (mcts_gpu.py:1410-1424).  `uniforms` float32[R] in [0,1) selects sampled picks from the
 * log-space stable policy (mcts_gpu.py:853-898) by inverse CDF; NULL = argmax (sample_moves=False).
 * Outputs are fully written: policy_dense float32[B,T] (0 fill), chosen_index int64[B] (-1),
 * chosen_codes int32[B,4] (-1), chosen_valid uint8[B] (0), root_value float32[R].  M <= 256. */
LZ_API int lz_root_finalize_from_visits(const int64_t* legal_index_mat, const int32_t* action_code_mat,
                                        const uint8_t* valid_mask, const float* visits,
                                        const float* value_sum, const int64_t* valid_root_indices,
                                        int64_t num_roots, int64_t max_actions, int64_t batch_size,
                                        int64_t total_action_dim, const float* root_temperatures,
                                        const float* uniforms, float* policy_dense,
                                        int64_t* chosen_index, int32_t* chosen_codes,
                                        uint8_t* chosen_valid, float* root_value, void* stream);

/* ---- self-play step operators --------------------------------------------------------------- */

/* Opening plies of a game (`opening_random_moves` of v1/python/self_play_gpu_runner.py; mcts_gpu.py:1425-1447 replaces
 * the search's pick by torch.multinomial over the uniform policy of the valid actions): for every packed row r whose root
 * valid_root_indices[r] (NULL: r itself) has force_mask[root] != 0, the chosen action becomes the k-th valid slot of the row
 * in ascending order, k = min(floor(uniforms[r] * n), n - 1), n = valid slots; chosen_valid_mask[root] = 1.  Rows without a
 * valid slot and unflagged roots are left alone; the policy target is not touched.  Run after
 * lz_root_finalize_from_visits on the same buffers. */
LZ_API int lz_root_force_uniform_picks(const int64_t* legal_index_mat, const int32_t* action_code_mat,
                                       const uint8_t* valid_mask, const int64_t* valid_root_indices, int64_t num_roots,
                                       int64_t max_actions, const uint8_t* force_mask, const float* uniforms,
                                       int64_t* chosen_action_indices, int32_t* chosen_action_codes,
                                       uint8_t* chosen_valid_mask, void* stream);

/* v0_core.self_play_step_inplace  (module.cpp:1387-1409, :632-871).  Mutates states / plies / done.
 * Per active row i it reports fin_kind[i] = 0 (game continues), 1 (ended before the move: terminal
 * root or no valid choice) or 2 (ended by the move: winner / draw / ply cap), with
 * result_from_black[i] and soft_value[i] = tanh(k*(black-white)/18). */
LZ_API int lz_self_play_step_inplace(const LzStateSoA* states, int64_t batch, int64_t* plies,
                                     uint8_t* done, const int64_t* active_idx, int64_t num_active,
                                     const int32_t* chosen_action_codes, const uint8_t* terminal_mask,
                                     const uint8_t* chosen_valid_mask, int64_t max_game_plies,
                                     float soft_value_k, int32_t* fin_kind, float* result_from_black,
                                     float* soft_value, void* stream);

/* v0_core.finalize_trajectory_inplace  (module.cpp:1410-1420, :547-630).
 * For each finished slot with step_counts > 0 writes value = sign*result, soft = sign*soft into the
 * target buffers at that game's step indices; keep[f] = step_counts[slot] > 0,
 * final_counts[f] = step_counts[slot]; counts_out int64[3] += [black wins, white wins, draws]
 * (caller zeroes counts_out). */
LZ_API int lz_finalize_trajectory_inplace(float* value_targets, float* soft_value_targets,
                                          const int8_t* player_signs, const int64_t* step_index_matrix,
                                          const int64_t* step_counts, int64_t num_games,
                                          int64_t max_steps, const int64_t* slots,
                                          const float* result_from_black,
                                          const float* soft_value_from_black, int64_t num_slots,
                                          uint8_t* keep, int64_t* final_counts, int64_t* counts_out,
                                          void* stream);


/* Fused per-ply tail of the v1 wave loop for a FIXED wave of G slots (v1/python/self_play_gpu_runner.py:205-247 with
 * v1/python/trajectory_buffer.py:63-140): finished slots stay in the batch and are masked by `done`, so that nothing
 * in the loop needs the host.  Same results as append_steps + step_index update + self_play_step_inplace +
 * finalize_trajectory_inplace on the live slots (tests/test_gpu_selfplay.py).
 *
 * lz_wave_record: live slot g (done[g]==0, ascending g) gets arena row *cursor + rank(g); its model input
 * float32[11*36], legal mask uint8[T], policy float32[T] are copied into the arena row, value / soft targets are set
 * to NaN, sign = (current_player >= 0 ? 1 : -1); step_index[g, step_counts[g]++] = row; rows[g] = row (-1 for
 * finished slots); *cursor += live count.  A row >= capacity or a full step_index row drops the sample and bumps
 * *overflow (the host sizes both so that this never happens).
 * cursor == NULL and step_index_matrix == NULL select the SLOT-MAJOR arena of the finished-row log
 * (lz_wave_log_finished): the row of slot g's step n is g * max_steps + n, capacity >= num_slots * max_steps. */
LZ_API int lz_wave_record(const uint8_t* done, int64_t num_slots, int64_t* cursor, int64_t capacity,
                          int64_t max_steps, int64_t* step_index_matrix, int64_t* step_counts, int64_t* rows,
                          int32_t* overflow, const float* model_input, const uint8_t* legal_mask,
                          const float* policy_dense, const int64_t* current_player, int64_t action_dim,
                          float* arena_state, uint8_t* arena_legal, float* arena_policy, float* arena_value,
                          float* arena_soft, int8_t* arena_sign, void* stream);

/* lz_wave_step_finish: for every live slot apply chosen_action_codes[g] (self_play_step_inplace rules,
 * module.cpp:724-856); a game that ends has value = sign*result / soft = sign*tanh(k*delta/18) written over its rows
 * (module.cpp:547-630), outcome int64[3] += [black wins, white wins, draws] (games with >= 1 recorded step),
 * delta_hist int64[37] (optional) += final black-white piece difference clamped to [-18,18], lengths[slot_game ?
 * slot_game[g] : g] (optional) = recorded steps, *finished (optional) += 1.  reseat == 0: done[g] = 1.  reseat != 0: the slot restarts from the empty
 * board (plies, step_counts = 0, reseated[g] = 1 if given) -- the steady-state population of bench.py.
 * step_index_matrix == NULL: slot-major arena, the rows of slot g are g * max_steps + [0, step_counts[g]). */
LZ_API int lz_wave_step_finish(const LzStateSoA* states, int64_t num_slots, int64_t* plies, uint8_t* done,
                               const int32_t* chosen_action_codes, const uint8_t* terminal_mask,
                               const uint8_t* chosen_valid_mask, int64_t max_game_plies, float soft_value_k,
                               float* value_targets, float* soft_value_targets, const int8_t* player_signs,
                               const int64_t* step_index_matrix, int64_t* step_counts, int64_t max_steps,
                               int64_t* outcome, int64_t* delta_hist, int64_t* lengths, const int64_t* slot_game,
                               int64_t* finished, uint8_t* reseated, int reseat, void* stream);

/* lz_wave_reseat: the wave loop of self_play_gpu_runner.py:84-90 starts the next `concurrent_games` games only when the
 * whole wave has finished; here finished slots (done[g] != 0, ascending g) restart from the empty board at once while
 * *budget (games not yet started) lasts: slot_game[g] = (*next_game)++, plies / step_counts = 0, done[g] = 0,
 * reseated[g] = 1 (optional); *budget is decremented.  Deterministic (one workgroup, ordered scan).
 * logged_only != 0: a finished slot that still holds rows (step_counts[g] > 0, not yet taken by
 * lz_wave_log_finished) is not re-seated. */
LZ_API int lz_wave_reseat(const LzStateSoA* states, int64_t num_slots, uint8_t* done, int64_t* plies,
                          int64_t* step_counts, int64_t* budget, int64_t* next_game, int64_t* slot_game,
                          uint8_t* reseated, int logged_only, void* stream);

/* lz_wave_log_finished: finished-row log of the streaming worker.  The reference worker sees the rows of a wave only
 * after the whole wave has drained and copies them to the host with the GPU idle
 * (v1/python/self_play_worker.py:430-546: `_run_once` -> `chunk_batch.to("cpu")` -> `save_self_play_payload`); here
 * the rows of a game leave the slot-major live arena (lz_wave_record with cursor == NULL) for a game-major log the
 * moment the game has ended, so finished samples can be copied out while the wave goes on playing.
 * For every slot with done[g] != 0 and step_counts[g] > 0, in ascending slot order: its step_counts[g] rows (state
 * float32[396], legal uint8[action_dim], policy float32[action_dim], value, soft value) are appended to the log at
 * log_state[0] while they fit log_capacity, and step_counts[g] = 0; slots that do not fit wait (lz_wave_reseat with
 * logged_only does not re-seat them).  log_state int64[4] = {rows in the log, games in the log, games waiting, rows
 * waiting}; log_base int64[num_slots] is scratch.  Deterministic (one-workgroup ordered scan, then one wave per slot). */
LZ_API int lz_wave_log_finished(const uint8_t* done, int64_t* step_counts, int64_t num_slots, int64_t max_steps,
                                int64_t action_dim, const float* arena_state, const uint8_t* arena_legal,
                                const float* arena_policy, const float* arena_value, const float* arena_soft,
                                float* log_state_rows, uint8_t* log_legal, float* log_policy, float* log_value,
                                float* log_soft, int64_t log_capacity, int64_t* log_state, int64_t* log_base,
                                void* stream);

/* ---- network forward ------------------------------------------------------------------------- */

/* Packed network description (built by liuzhou_amd/net_pack.py from a ChessNet state_dict:
 * BatchNorm folded, conv weights in v_mfma_f32_16x16x32_f16 operand order).  All offsets into
 * `fparams` are in floats, `layer_offsets` (stem, conv1/conv2 per block, stacked head convs) in halfs. */
#define LZ_NET_MAX_LAYERS 96     /* stem + 2 convs per block + the stacked head convs: up to 47 residual blocks */
typedef struct LzNetDesc {
    int32_t channels;            /* trunk channels: 64 or 128 */
    int32_t blocks;              /* residual blocks (<= (LZ_NET_MAX_LAYERS - 2) / 2 = 47) */
    int32_t num_layers;          /* 2 + 2*blocks */
    int32_t max_blocks;          /* persistent grid size (0 = 256, one workgroup per CU) */
    const void* wfrag;           /* device, fp16 */
    const float* fparams;        /* device, fp32 */
    int64_t wfrag_bytes;         /* sizes of the two buffers (bounds of the kernel's buffer descriptors) */
    int64_t fparams_bytes;
    int32_t layer_offsets[LZ_NET_MAX_LAYERS];
    int32_t head_frag_offsets[4]; /* halfs: gpool_linear [64x192], fc1 [128x192], fc2 [112x128], out convs [16x64] */
    int32_t off_stem_bias, off_block0 /* a1|b1|bias1 per block, 3*C floats each */, off_trunk_a, off_trunk_b,
            off_head_bias, off_p_gwT, off_p_a2, off_p_b2, off_p_out, off_v_w1T, off_v_b1, off_v_w2T, off_v_b2;
    int32_t flags;               /* bit 0 (64 channels): 4-wave workgroups of 8 samples, two per CU (default grid 512):
                                    a half-size batch then still covers every CU -- for two half-batches evaluated
                                    concurrently on two streams;
                                    bit 1 (128 channels): 4-wave workgroups with 4 channel tiles per wave (one wave per
                                    SIMD, half the LDS operand reads; measured 2 % slower, off by default);
                                    bit 2: fp32 OPERANDS (parity mode, csrc/lz_net_f32.hip: v_mfma_f32_16x16x4_f32 on
                                    `wfrag_f32`): the reference's fp32 forward within 1e-5, at a fraction of the speed;
                                    every lz_net_forward_* entry point and the search loops honour it;
                                    bit 3: SPLIT fp16 operands (round 6, same file): every conv operand as hi + lo * 2^-11
                                    (two fp16 numbers, 22 bits), every product as three v_mfma_f32_16x16x32_f16 -- the
                                    fp32 forward within 1e-5 like bit 2, at 3/16 of its matrix-pipe time; needs `wfrag_lo` */
    const float* wfrag_f32;      /* device, fp32 conv fragments [layer][tap][K/4][Cout/16][64 lanes] at the same element
                                    offsets as `wfrag` (layer_offsets); NULL unless flags bit 2 is used */
    int64_t wfrag_f32_bytes;
    const void* wfrag_lo;        /* device, fp16: the LOW halves of the conv weights, fp16((w - fp16(w)) * 2^11), in the
                                    order and at the offsets of `wfrag` (conv layers only); NULL unless flags bit 3 is used */
    int64_t wfrag_lo_bytes;
} LzNetDesc;

/* ChessNet.forward (src/neural_network.py:213-259) + bucket_logits_to_scalar (:201-210), fused:
 * planes float32[N,11,6,6] -> log_p1 / log_p2 / log_pmc float32[N,36] (log-softmax over the board),
 * value_logits float32[N,101] (may be NULL), value float32[N] = E[bucket centre] (may be NULL).
 * fp16 MFMA operands with fp32 accumulation and an fp32 residual stream (the reference's autocast
 * inference mode, v1/python/mcts_gpu.py:640-646). */
LZ_API int lz_net_forward_f16(const LzNetDesc* net, const float* planes, int64_t batch,
                              float* log_p1, float* log_p2, float* log_pmc,
                              float* value_logits, float* value, void* stream);
/* same network, input staged directly from 32-byte packed bitboard states (lz_pack_states) instead of
 * float planes: the model-input encode (v0/src/net/encoding.cpp:26-79) is fused into the kernel's prologue */
LZ_API int lz_net_forward_packed_f16(const LzNetDesc* net, const void* packed_states, int64_t batch,
                                     float* log_p1, float* log_p2, float* log_pmc,
                                     float* value_logits, float* value, void* stream);
/* In both entry points log_p1 / log_p2 / log_pmc may all be NULL (values only: the policy head is skipped; `value`
 * is then required).  The `_counted` form takes the batch size from device memory (`*count`, clamped to `capacity`),
 * so that a producer kernel can size the batch without a host round trip (fused root search, lz_root_prepare). */
LZ_API int lz_net_forward_packed_counted_f16(const LzNetDesc* net, const void* packed_states, int64_t capacity,
                                             const int64_t* count, float* log_p1, float* log_p2, float* log_pmc,
                                             float* value_logits, float* value, void* stream);
/* one-time kernel attribute setup (dynamic LDS size); call once per process before graph capture */
LZ_API int lz_net_configure(void);
/* sizeof(LzNetDesc) as the library was compiled: a binding that lays the struct out itself (ctypes) compares */
LZ_API int64_t lz_net_desc_bytes(void);
/* measurement aid (bench.py): when enabled every lz_net_forward_f16 launch is bracketed by HIP events on
 * its own stream; after synchronising, lz_prof_net_summary returns the summed kernel time. */
LZ_API int lz_prof_enable(int on);
LZ_API int lz_prof_net_summary(double* total_ms, int64_t* launches, int64_t* evals);
/* launches on several streams may overlap: time during which at least one bracketed launch was running */
LZ_API int lz_prof_net_busy(double* busy_ms);
/* the HBM-bound kernels of the tree search, bracketed the same way while lz_prof_enable(1): kind 0 = the fused
 * expand + backup + select kernel of one simulation (units = games), kind 1 = lz_tree_advance (units = games) */
LZ_API int lz_prof_aux_summary(int kind, double* total_ms, int64_t* launches, int64_t* units);

/* ---- device-resident tree search (variant P) --------------------------------------------------- */

/* Replaces the reference's portable full-tree search and its split-phase C++ twin:
 *   v1/python/portable_mcts.py:264-746 (PortableMCTS.search_batch) == src/mcts.py:280-548 with batch_K=1,
 *   v1/cpp/portable_mcts.cpp:448-979 (PrepareRoots / SelectLeaves / CompletePending / AdvanceRoots).
 * All buffers are caller-allocated device memory.  Per game g:
 *   nodes  [g*node_cap  .. +node_cap)   node_cap >= sims + 2 (+ the kept subtree with lz_tree_advance), <= 524288 (lz_tree_advance: four
 *                                       waves per workgroup up to 65 536 nodes, one above)
 *   path   [g*path_cap  .. +path_cap)   path_cap >= 3; a descent stops at path_cap - 1 levels (a game lasts <= 144
 *                                       plies, so 160 entries hold every reachable path)
 * Per engine: ONE edge pool `edges` of pool_chunks x edge_chunk 32-byte records, handed to the games in chunks
 * (round 4; it replaces per-game regions of (sims + 1) * 72 edges, the worst case no game ever reaches):
 *   chunk_list [g*chunk_cap .. +chunk_cap)  ids of the chunks game g owns, in the order it took them; n_chunks[g]
 *   free_chunks[pool_chunks], pool_top[1]   stack of free chunk ids and its height; the caller initialises them to
 *                                           0 .. pool_chunks-1 and pool_chunks, n_chunks / n_edges to 0
 *   pool_stats[3]   [0] += expansions refused because no chunk was free (the leaf stays unexpanded, its value is still
 *                   backed up, the next visit tries again -- results then differ from an unbounded tree, so size the
 *                   pool so that this stays 0); [1] = min(free chunks seen) (initialise to pool_chunks); [2] = fresh roots
 *                   that have not taken the chunk of their first expansion yet (initialise to 0): a chunk stays reserved
 *                   for each of them, so pool_chunks >= num_games is required
 * Every edge index in a record or hand-off array (node edge_begin, edge cbegin, path, leaf_edge) is an index into the
 * pool; (pool_chunks + 1) * edge_chunk <= 2^31.  chunk_cap * (edge_chunk - 71) >= node_cap * 72 makes the node arena the
 * only per-game bound (required by lz_tree_advance).  A kept subtree that would not leave room for the next search in
 * the NODE arena is pruned and counted (lz_tree_advance).
 * States are 32-byte packed bitboard records (lz_pack_states). */
typedef struct LzTreeDesc {
    int64_t num_games;
    int32_t node_cap, edge_chunk /* edges per chunk: power of two >= 128 */, path_cap, chunk_cap;
    double  exploration_weight;
    const void* root_state;        /* packed [B]: current game states (input of lz_tree_begin) */
    void*    nodes;                /* [B*node_cap] 48-byte records {packed state, int32 edge_begin, int32 nedges (-1 =
                                      not expanded), int32 parent node (-1 root), 4 B pad} */
    void*    edges;                /* [pool_chunks*edge_chunk] 32-byte records {double W (value sum, child mover's view),
                                      float P, uint32 N | info<<24, int32 child node or -1, int32 child edge_begin (pool
                                      index), uint8 action, uint8 child nedges, 6 B pad};
                                      info: bit0 child mover white, bit1 terminal, bits2-3 terminal value + 1 */
    int32_t* n_nodes;              /* [B] */
    int32_t* n_edges;              /* [B] next free pool index in the game's open chunk (multiple of edge_chunk: none) */
    int32_t* root_visits;          /* [B] */
    double*  root_w;               /* [B] */
    float*   root_init_value;      /* [B] */
    int32_t* path;                 /* [B*path_cap] edge index | bit 31: the mover changes from parent to child */
    int32_t* path_len;             /* [B] */
    int32_t* leaf_kind;            /* [B] 0 inactive, 1 needs evaluation, 2 terminal (value in leaf_value),
                                      3 root kept by lz_tree_advance (no evaluation, noise mix only) */
    void*    leaf_state;           /* packed [B]: state awaiting evaluation */
    float*   leaf_value;           /* [B] */
    uint8_t* root_terminal;        /* [B] */
    const uint8_t* active;         /* [B] or NULL (all active) */
    int32_t* leaf_edge;            /* [B] edge the pending leaf hangs from (select -> expand hand-off) */
    int32_t* leaf_parent;          /* [B] node that owns that edge */
    /* Optional per-simulation trace of what the expand step consumed (parity tests of the production launch path:
     * lz_tree_search / lz_tree_search_continue under a hipGraph).  Slot 0 is the root step, slot s the s-th simulation;
     * steps beyond trace_cap are not recorded.  All NULL / 0 in production. */
    int32_t* trace_kind;           /* [trace_cap][B] leaf_kind the step saw */
    void*    trace_leaf;           /* [trace_cap][B] packed state that was evaluated */
    float*   trace_heads;          /* [trace_cap][B][108] log_p1 | log_p2 | log_pmc rows the step read (heads mode) */
    float*   trace_priors;         /* [trace_cap][B][220] softmax over the legal set before noise / renormalisation */
    float*   trace_value;          /* [trace_cap][B] evaluator value the step read */
    int64_t  trace_cap;
    int32_t* eval_count;           /* optional [B]: += 1 for every evaluation a game's expand step CONSUMED (a leaf or root it
                                    * expanded); terminal leaves, inactive slots and kept roots do not count.  NULL: off */
    int32_t* chunk_list;           /* [B*chunk_cap] */
    int32_t* n_chunks;             /* [B] */
    int32_t* free_chunks;          /* [pool_chunks] */
    int32_t* pool_top;             /* [1] */
    int32_t* pool_stats;           /* [3] */
    int64_t  pool_chunks;
    /* optional compact evaluation list of lz_tree_search / lz_tree_search_continue (all three set, live_count_cap >= sims + 2):
     * every simulation's leaves that NEED the network (live game, leaf to expand -- not a terminal leaf, not a kept
     * root) are appended to live_state (<= num_games records) and evaluated with lz_net_forward_packed_counted_f16, so a
     * launch runs ceil(live / samples-per-pass) network passes instead of one per slot: the cost of a wave that is
     * draining follows its live games (the reference's PortableMCTS.evaluate_states gets exactly the pending leaves,
     * v1/python/portable_mcts.py:337-378).  live_row[g] = row of game g's leaf in the list; live_count[s] = number
     * of leaves of simulation s.  Results are bit-identical to the dense launch. */
    void*    live_state;           /* [num_games] 32-byte packed states */
    int32_t* live_row;             /* [num_games] */
    int64_t* live_count;           /* [live_count_cap] */
    int64_t  live_count_cap;
} LzTreeDesc;

/* SoA batch -> packed records; packed records -> float32[B,11,6,6] model input (src/neural_network.py:15-65) */
/* Wave-batched leaves: the legacy search of src/mcts.py:280-497 (`batch_K` distinct leaves per tree and wave, no
 * virtual loss; batch_K = 1 is the protocol above).  Per-leaf arrays are slot-major [batch_k][B]; `leaf_state` is the
 * leaf of slot (j, g); the leaves that need the network are also appended to a compact list (`eval_state`, `eval_count`)
 * whose rows the network evaluates (device-counted batch) and lz_tree_wave_expand reads back through `eval_row`. */
typedef struct LzTreeWaveDesc {
    int32_t  batch_k;              /* leaves per game and wave, 1..32 */
    int32_t  path_cap;             /* entries per leaf path, > 160 (the level stack of a walk; a game lasts <= 144 plies) */
    int32_t* path;                 /* [batch_k][B][path_cap] */
    int32_t* path_len;             /* [batch_k][B] */
    int32_t* leaf_kind;            /* [batch_k][B] 0 inactive, 1 needs evaluation, 2 terminal */
    void*    leaf_state;           /* [batch_k][B] packed */
    float*   leaf_value;           /* [batch_k][B] */
    int32_t* leaf_edge;            /* [batch_k][B] */
    int32_t* leaf_parent;          /* [batch_k][B] */
    int32_t* sims_done;            /* [B] simulations used by the current search */
    int32_t* unfinished;           /* [1] games with budget left after the last lz_tree_wave_select */
    int32_t* eval_row;             /* [batch_k][B] row of the slot's leaf in the compact evaluation list */
    void*    eval_state;           /* [batch_k * B] packed: the leaves of the wave that need the network, compacted */
    int64_t* eval_count;           /* [1] length of that list (device-side batch size of the network launch) */
    int64_t* eval_total;           /* [1] sum of eval_count over the previous waves (statistics) */
    int32_t  max_backtrack_steps;  /* the reference's MAX_BACKTRACK_STEPS (src/mcts.py:337): a walk gives up after this many
                                      upward moves; every walk of a wave restarts at the root and replays the earlier
                                      ones, so the count is cumulative over the wave and the wave ends there.  0 = 128 */
    int32_t  reserved_;
} LzTreeWaveDesc;
/* SelectLeaves for a wave: up to min(batch_k, sims - sims_done) distinct leaves per game, in the order the reference
 * collects them (src/mcts.py:333-425); sims_done += leaves found.  reset_budget != 0 starts a new search. */
LZ_API int lz_tree_wave_select(const LzTreeDesc* tree, const LzTreeWaveDesc* wave, int64_t sims, int reset_budget,
                               void* stream);
/* CompletePending for a wave (src/mcts.py:427-497): terminal / no-legal-move leaves are backed up first, then the
 * evaluated leaves are expanded and backed up, each group in leaf order.  Evaluator rows (heads or priors220, values)
 * are the rows of the compact list, or slot-major [batch_k][B] when `slot_major` != 0 or priors220 is given. */
LZ_API int lz_tree_wave_expand(const LzTreeDesc* tree, const LzTreeWaveDesc* wave, const float* log_p1,
                               const float* log_p2, const float* log_pmc, const float* priors220, const float* values,
                               int slot_major, void* stream);
/* Whole search of one move in waves, enqueued from C++ (hipGraph-capturable); see lz_engine.hip. */
LZ_API int lz_tree_search_waves(const LzTreeDesc* tree, const LzTreeWaveDesc* wave, const LzNetDesc* net, int64_t sims,
                                int64_t waves, float* log_p1, float* log_p2, float* log_pmc, float* values,
                                const float* noise, int64_t noise_stride, float epsilon, int continue_trees,
                                int skip_roots, void* stream);

LZ_API int lz_pack_states(const LzStateSoA* states, int64_t batch, void* packed_out, void* stream);
LZ_API int lz_packed_to_model_input(const void* packed, int64_t batch, float* out, void* stream);

/* PrepareRoots: fresh single-node trees from root_state; the roots become the pending evaluations. */
LZ_API int lz_tree_begin(const LzTreeDesc* tree, void* stream);
/* SelectLeaves: one PUCT descent per game; leaf_kind / leaf_state / path are written. */
LZ_API int lz_tree_select(const LzTreeDesc* tree, void* stream);
/* CompletePending: expand the pending leaf of every game and back the value up (is_root: expand only,
 * optional noise float32[B,noise_stride] mixed with weight epsilon, `set_root_priors` semantics).
 * Evaluator output is either the three log-prob heads float32[B,36] (priors220 == NULL) or a dense
 * prior row float32[B,220] (injected evaluator for parity runs); values float32[B]. */
LZ_API int lz_tree_expand(const LzTreeDesc* tree, int is_root, const float* log_p1, const float* log_p2,
                          const float* log_pmc, const float* priors220, const float* values,
                          const float* noise, int64_t noise_stride, float epsilon, void* stream);
/* Root policy and move pick (portable_mcts.py:150-261, :690-727), per-child statistics (child_* are [B,out_cap]).
 *   selection policy = visits^(1/T) in log space with `temperatures`; it drives the move: sample_moves != 0:
 *     inverse-CDF sample with uniforms[g]; 0: most visits -> Q -> prior -> lowest index; force_uniform[g] != 0
 *     (opening plies): uniform over the legal children, index floor(uniforms[g] * n).
 *   policy_dense (training target) = the same with `target_temperatures` (NULL: = temperatures) on the scores
 *     visits + prior_pseudocount * normalised priors  (policy_target_temperature / _prior_pseudocount). */
LZ_API int lz_tree_finish(const LzTreeDesc* tree, const float* temperatures, const float* target_temperatures,
                          float prior_pseudocount, const uint8_t* force_uniform, int sample_moves,
                          const float* uniforms,
                          float* policy_dense /*[B,220]*/, int32_t* chosen_index, int32_t* chosen_code /*[B,4]*/,
                          uint8_t* chosen_valid, uint8_t* terminal_mask, float* root_value,
                          int32_t* child_count, int32_t* child_action, int32_t* child_visits,
                          float* child_prior, int64_t out_cap, void* stream);
/* One whole search enqueued from C++: begin, root evaluation + expansion, then `sims` x
 * (select -> planes -> fused network -> expand + backup).  No host synchronisation; capturable. */
LZ_API int lz_tree_search(const LzTreeDesc* tree, const LzNetDesc* net, int64_t sims, float* planes /*[B,11,36]*/,
                          float* log_p1, float* log_p2, float* log_pmc, float* values, const float* noise,
                          int64_t noise_stride, float epsilon, void* stream);
/* ---- per-game counter RNG -----------------------------------------------------------------------------------
 * Replaces the library generators behind the reference's root noise and move sampling
 * (torch.distributions.Gamma / torch.multinomial on the device generator, v1/python/mcts_gpu.py:1329-1339,1410-1424;
 * torch Dirichlet in v1/python/portable_mcts.py:302-317; np.random.dirichlet in src/mcts.py:488-491), whose streams
 * depend on slot and batch composition.  Philox4x32-10 keyed by `seed`, counter = (game id, ply, purpose, index):
 * every variate is a pure function of those, so a game plays the same moves whichever slot / stream / rank runs it.
 *   lz_rng_gamma:   out[g*stride + k] = Gamma(alpha, 1) draw k of game_id[g] at ply[g], k < count (Marsaglia-Tsang);
 *                   normalised over a game's legal children these are Dirichlet(alpha) noise.
 *   lz_rng_uniform: out[g] = uniform [0,1) for `purpose` (1 = move pick, 2 = opening move) of game_id[g] at ply[g].
 * game_id NULL: g itself; ply NULL: 0. */
LZ_API int lz_rng_gamma(uint64_t seed, const int64_t* game_id, const int64_t* ply, int64_t batch, float alpha,
                        int64_t count, float* out, int64_t stride, void* stream);
LZ_API int lz_rng_uniform(uint64_t seed, const int64_t* game_id, const int64_t* ply, int64_t batch, int purpose,
                          float* out, void* stream);

/* ---- fused root-PUCT search (variant R) on packed states -------------------------------------------------
 * The host chain of v1/python/mcts_gpu.py:1249-1457 (encode -> project -> root_pack -> noise -> apply -> evaluate
 * children -> perspective / terminal / soft value -> leaf matrix) as two fixed-shape kernels around the network
 * launches; afterwards lz_root_puct_allocate_visits / lz_root_finalize_from_visits run on the padded [B,72] rows
 * (valid_root_indices = 0..B-1).  No host synchronisation anywhere: the whole search is graph-capturable.
 *   lz_root_prepare: per root -- legal set (tensor semantics), masked softmax of the combined head logits, rows packed
 *     left to 72 slots (legal_index_mat int64, priors_mat renormalised, action_code_mat int32x4, valid_mask, counts,
 *     terminal_mask = no legal action), optional Dirichlet mix (noise float32[B,72], per-row normalised, weight
 *     epsilon, rows with > 1 action), leaf_mat zeroed, child states appended to child_states (packed records) with
 *     child_ref = root*72 + slot; *n_children (device uint64, reset by the call) = number of children.
 *   lz_root_collect: child values (network, child mover's view) -> leaf_mat in the parent's view; children whose
 *     game is over get tanh(k * (black - white) / 18) from the parent's side instead. */
LZ_API int lz_root_prepare(const void* root_states, int64_t batch, const float* log_p1, const float* log_p2,
                           const float* log_pmc, const float* noise, float epsilon, int64_t* legal_index_mat,
                           float* priors_mat, int32_t* action_code_mat, uint8_t* valid_mask, int32_t* counts,
                           uint8_t* terminal_mask, float* leaf_mat, void* child_states, int32_t* child_ref,
                           uint64_t* n_children, int64_t child_capacity /* records in child_states / child_ref,
                           >= 72 * batch */, int32_t* overflow /* device counter of roots dropped because the list
                           was full (0 in correct use), may be NULL */, void* stream);
LZ_API int lz_root_collect(const void* root_states, const void* child_states, const int32_t* child_ref,
                           const float* child_values, const uint64_t* n_children, int64_t capacity,
                           float soft_value_k, float* leaf_mat, void* stream);
/* Top-K lookahead of the root search (`sparse_ply` > 1: V1RootMCTS._refine_via_topk_lookahead,
 * v1/python/mcts_gpu.py:976-1046, driven from :1150-1160), as two fixed-shape stages around one more round of
 * network + lz_root_prepare + network + lz_root_collect on the batch * top_k L2 positions:
 *   lz_root_topk_children: per root the top_k VALID children by leaf value (highest first, lowest slot among equals)
 *     -> top_slot int32[batch, top_k] (-1: fewer legal actions than that) and l2_states (packed records,
 *     batch * top_k; the position after that action; an all-zero record -- phase 0, no legal action, so that
 *     lz_root_prepare gives it no children -- for the empty picks).
 *   lz_root_refine_topk: leaf_mat[b][top_slot[b][k]] = max(itself, max over the valid entries of row b*top_k + k of
 *     the L2 leaf matrix (0 when that row has no valid entry or a non-finite maximum)).
 * A picked child without a legal reply has no grandchildren: its lookahead value is 0, as for a row whose maximum is
 * not finite (the reference's reshape at mcts_gpu.py:1030 fails on a batch that holds such a child; the operator chain
 * of liuzhou_amd/mcts_gpu.py defines it this way and the two agree). */
LZ_API int lz_root_topk_children(const void* root_states, int64_t batch, const float* leaf_mat,
                                 const uint8_t* valid_mask, const int32_t* action_code_mat, int64_t top_k,
                                 int32_t* top_slot, void* l2_states, void* stream);
LZ_API int lz_root_refine_topk(int64_t batch, int64_t top_k, const int32_t* top_slot, const float* l2_leaf_mat,
                               const uint8_t* l2_valid_mask, float* leaf_mat, void* stream);

/* AdvanceRoots (src/mcts.py:577-592, portable_mcts.py:74-87, portable_mcts.cpp:739-769): after the host has
 * played `played_action[g]` (220-d index, -1: none) and refreshed root_state, promote that child to root and keep
 * its subtree (compacted in place) with its statistics.  Games with reset[g] != 0, inactive games, children that
 * were never expanded or a child state different from root_state[g] start a fresh tree instead.  The reference's tree
 * is unbounded; here a kept subtree that would leave no room for `next_sims` more simulations in the game's NODE arena
 * is PRUNED to its oldest part that fits (a prefix in expansion order -- closed under "parent of"; edges whose child fell
 * past the cut keep their statistics and are expanded afresh when visited).  Edge room is pooled per engine and does not
 * bound a single game.  `dropped` / `pruned`: device int32[1] each, may be NULL: += subtrees forgotten whole (defensive:
 * only if not even the first 64-node chunk fits) / += subtrees pruned.
 * Follow with lz_tree_search_continue (or lz_tree_expand(is_root=1) + the split-phase loop): kept roots are not
 * re-evaluated, they only get a fresh noise mix (portable_mcts.py:617-621). */
LZ_API int lz_tree_advance(const LzTreeDesc* tree, const int32_t* played_action, const uint8_t* reset,
                           int64_t next_sims, int32_t* dropped, int32_t* pruned, void* stream);
/* measurement aid (scripts/exp_advance_phases.py): when set, every later lz_tree_advance writes per game int64[8] =
 * 100 MHz ticks at its start / after the marks / after the nodes / after the edge runs, nodes before, nodes kept, 0, 0
 * (games that start a fresh tree only write the first and the fifth).  NULL (the default) switches it off. */
LZ_API int lz_debug_advance_ticks(int64_t* ticks);
/* lz_tree_search without the begin: searches the trees prepared by lz_tree_advance. */
LZ_API int lz_tree_search_continue(const LzTreeDesc* tree, const LzNetDesc* net, int64_t sims, float* planes,
                                   float* log_p1, float* log_p2, float* log_pmc, float* values, const float* noise,
                                   int64_t noise_stride, float epsilon, void* stream);

/* The same search (fresh: continue_trees = 0, or on the trees prepared by lz_tree_advance: 1) as ONE kernel launch in
 * which a workgroup owns 8 games for the whole move: per simulation one network pass on its 8 pending leaves, then the
 * expand / backup / select step of those games, with only a workgroup barrier in between -- no kernel boundary and no
 * device-wide barrier per simulation (replaces the host loop of v1/cpp/portable_mcts.cpp:483-590,832-939 /
 * v1/python/portable_cpp_mcts.py:270-282 like lz_tree_search does; identical results: the same device code runs the
 * network pass and the tree steps).  Two such workgroups share a CU, so one's tree step overlaps the other's network
 * pass.  64-channel networks in fp16 mode only: anything else returns LZ_ERR_UNSUPPORTED and the caller uses
 * lz_tree_search.
 *   cu_slots:    int32[4096] scratch (zeroed by the call) or NULL; with stagger_us > 0 the second workgroup to arrive
 *                on a CU starts stagger_us microseconds late, so that the pair alternates its phases;
 *   phase_ticks: optional int64[grid][4] = 100 MHz ticks each workgroup spent in network passes / tree steps, its
 *                arrival slot on its CU, the CU key (grid = lz_tree_search_persistent_grid(num_games)); NULL in production. */
LZ_API int lz_tree_search_persistent(const LzTreeDesc* tree, const LzNetDesc* net, int64_t sims, float* log_p1,
                                     float* log_p2, float* log_pmc, float* values, const float* noise,
                                     int64_t noise_stride, float epsilon, int continue_trees, int32_t* cu_slots,
                                     int64_t stagger_us, int64_t* phase_ticks, void* stream);
LZ_API int lz_tree_search_persistent_grid(int64_t num_games);

/* ---- training loss (the step right after the path) --------------------- */

/* Fused forward + backward of the reference's training loss (v1/python/train_bridge.py:330-375):
 *   policy: build_combined_logits -> masked_log_softmax -> batched_policy_loss (src/policy_batch.py:95-189),
 *   value : two-hot bucket cross entropy on clamp((1-alpha)*value + alpha*soft, -1, 1) (src/neural_network.py:176-198),
 *   WDL auxiliary term reported only (its weight is 0 in the reference).
 * Inputs float32: head outputs log_p1/log_p2/log_pmc [B,36], value_logits [B,101]; legal_mask uint8 [B,220];
 * policy_target [B,220]; value_target / soft_value_target [B]; policy_weight_sum = device scalar
 * sum_b (|value_b| < 1e-8 ? policy_draw_weight : 1).
 * Outputs: terms [B,4] = {KL_b, weight_b, bucket CE_b, WDL aux_b}  (loss = sum(KL*w)/(sum(w)+1e-8) + mean(CE));
 * gradients of grad_scale * loss with respect to the four head outputs (same shapes as the inputs). */
LZ_API int lz_policy_value_loss_fwd_bwd(const float* log_p1, const float* log_p2, const float* log_pmc,
                                        const float* value_logits, const uint8_t* legal_mask,
                                        const float* policy_target, const float* value_target,
                                        const float* soft_value_target, int64_t batch, float soft_label_alpha,
                                        float anti_draw_penalty, float policy_draw_weight,
                                        const float* policy_weight_sum, float grad_scale, float* terms,
                                        float* grad_log_p1, float* grad_log_p2, float* grad_log_pmc,
                                        float* grad_value_logits, void* stream);

/* ---- compact trajectory records: wire format of the per-iteration gather ---- */

/* A row of the 5-tensor trajectory contract (v1/python/trajectory_buffer.py:11-33; 2 692 B) as an exact 360-byte
 * record: u64[4] {own | phase<<36, opp, own-marked, opp-marked}, u32[7] legal bits, f32[72] policy of the legal
 * actions in ascending index order, f32 value, f32 soft value, 4 B pad.  unpack(pack(x)) reproduces every byte.
 * `not_representable` (device int32, add-only) counts rows whose planes are not 0/1, whose policy is non-zero off the
 * legal set, or that have more than 72 legal actions -- never the case for self-play output. */
#define LZ_TRAJECTORY_RECORD_BYTES 360
LZ_API int lz_pack_trajectory_rows(const float* state_tensors, const uint8_t* legal_masks,
                                   const float* policy_targets, const float* value_targets,
                                   const float* soft_value_targets, int64_t rows, void* records,
                                   int32_t* not_representable, void* stream);
LZ_API int lz_unpack_trajectory_rows(const void* records, int64_t rows, float* state_tensors, uint8_t* legal_masks,
                                     float* policy_targets, float* value_targets, float* soft_value_targets,
                                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIUZHOU_HIP_H */
