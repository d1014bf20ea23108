// lz_engine.hip -- device-resident full-tree PUCT search ("variant P"), one wavefront per game.
//
// Semantics follow v1/python/portable_mcts.py:264-746 (== src/mcts.py:280-548 with batch_K = 1, no virtual
// loss): one leaf per tree per simulation, all trees of the wave advanced together;
//   select : argmax_a Q + c*P*sqrt(max(1,N_parent))/(1+n_a) in double, Q = W/n flipped iff the mover
//            changes, first (lowest action index) maximum wins          (portable_mcts.py:480-506)
//   expand : children in ascending 220-d action index, priors renormalised over the legal set,
//            game-over children / no-legal leaves are terminal            (portable_mcts.py:418-478)
//   backup : leaf -> root, N += 1, W += v, v = -v iff the mover changes   (portable_mcts.py:123-138)
//   root   : expanded without backup, optional Dirichlet mix on the priors (portable_mcts.py:451-463)
//
// Layout in HBM:
//   per game g: node arena  [node_cap] x 48 B  { packed 32-byte state, edge_begin, n_edges, parent }
//               path        [path_cap] edge ids of the current simulation
//   per engine: edge pool   [pool_chunks x chunk] x 32 B { W f64, P f32, N|info u32, child i32, child edges i32+u8,
//               action u8 }, handed to the games in chunks (lz_tree_dev.h) -- sized for the mean fan-out, not for 72
//               children per node; a refused allocation is counted, never a fault
// A wave owns one game: lanes enumerate legal actions with ballots + popcount prefixes, evaluate PUCT
// scores for up to 2 children per lane and reduce with shuffles; there is no cross-wave communication.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <math.h>
#include <stdint.h>

#include <mutex>

#include "lz_rng.h"
#include "lz_tree_dev.h"

namespace {

// ---- SoA <-> packed -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pack_states_kernel(LzStateSoA s, int64_t B, Packed* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    State st;
    st.black = cells_equal(s.board + i * 36, 1);
    st.white = cells_equal(s.board + i * 36, -1);
    st.mb = cells_nonzero(s.marks_black + i * 36);
    st.mw = cells_nonzero(s.marks_white + i * 36);
    st.phase = (int)s.phase[i]; st.player = (int)s.current_player[i];
    st.pm_req = (int)s.pending_marks_required[i]; st.pm_rem = (int)s.pending_marks_remaining[i];
    st.pc_req = (int)s.pending_captures_required[i]; st.pc_rem = (int)s.pending_captures_remaining[i];
    st.forced = (int)s.forced_removals_done[i]; st.move_count = (int)s.move_count[i];
    st.msc = (int)s.moves_since_capture[i];
    out[i] = pack(st);
}

// packed states -> float32 model input planes (one wave per state; src/neural_network.py:15-65)
__global__ __launch_bounds__(kBlock) void packed_planes_kernel(const Packed* __restrict__ states, int64_t B,
                                                               float* __restrict__ out) {
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= B) return;
    const State s = unpack(states[g]);
    const bool blk = s.player == 1;
    const uint64_t own = blk ? s.black : s.white, opp = blk ? s.white : s.black;
    const uint64_t sm = blk ? s.mb : s.mw, om = blk ? s.mw : s.mb;
    float4* orow = reinterpret_cast<float4*>(out + g * 396);
    for (int j = lane; j < 99; j += kWave) {
        const int plane = j / 9, cell0 = (j - plane * 9) * 4;
        uint32_t bits;
        if (plane < 4) {
            const uint64_t src = plane == 0 ? own : plane == 1 ? opp : plane == 2 ? sm : om;
            bits = (uint32_t)(src >> cell0) & 0xFu;
        } else bits = (s.phase == plane - 3) ? 0xFu : 0u;
        orow[j] = make_float4((bits & 1) ? 1.f : 0.f, (bits & 2) ? 1.f : 0.f, (bits & 4) ? 1.f : 0.f, (bits & 8) ? 1.f : 0.f);
    }
}

__global__ __launch_bounds__(kBlock) void tree_begin_kernel(Tree t) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= t.B) return;
    begin_game(t, g);
}


__global__ __launch_bounds__(kBlock) void tree_select_kernel(Tree t) {
    const int g = wave_game();
    if (g >= t.B) return;
    tree_select(t, g, lane_id(), load_root_info(t, g));
}


// COMPACT: the evaluator rows are those of the compact list (Tree::live_row); tree_expand indexes its inputs with g, so
// the base pointers are shifted by row - g (rows of leaves that needed no evaluation are never read for their content)
template <bool IS_ROOT, bool COMPACT = false>
__global__ __launch_bounds__(kBlock) void tree_expand_kernel(Tree t, const float* __restrict__ lp1,
                                                             const float* __restrict__ lp2,
                                                             const float* __restrict__ lpm,
                                                             const float* __restrict__ priors220,
                                                             const float* __restrict__ values,
                                                             const float* __restrict__ noise, int noise_stride,
                                                             float epsilon, int step) {
    LZ_EXPAND_SCRATCH(sc);
    const int g = wave_game();
    if (g >= t.B) return;
    ptrdiff_t o = 0;
    if (COMPACT) o = (ptrdiff_t)(t.leaf_kind[g] == kLeafExpand ? t.live_row[g] : 0) - g;
    tree_expand<IS_ROOT>(t, g, lane_id(), lp1 ? lp1 + o * 36 : nullptr, lp2 ? lp2 + o * 36 : nullptr,
                         lpm ? lpm + o * 36 : nullptr, priors220 ? priors220 + o * 220 : nullptr, values + o, noise,
                         noise_stride, epsilon, sc, nullptr, step);
}

// The leaves that need the network (leaf_kind == kLeafExpand: fresh roots, leaves to expand -- not terminal leaves, kept
// roots, finished games) -> the compact list of the next network launch: live_row[g] = row of game g's leaf,
// live_state[row] = leaf_state[g], *count = how many.  ONE workgroup, one scan over per-thread counts.  (A first version let every select wave append with one atomicAdd on a shared
// counter: 16 384 same-address atomics serialise at ~9.5 ns each -- the tree kernel went from 63 to 219 us per simulation at
// C3, profiles/r05_experiments.md section 2.)
constexpr int kScanBlock = 1024;
constexpr int kScanUnroll = 8;
// Thread t looks at games t, t + 1024, t + 2048, ... (coalesced rounds; every load of a group of 8 rounds is in flight
// before the first is used -- a first version that gave each thread 16 CONSECUTIVE games and walked them one dependent
// load after the other took 54 us at 16 384 games) and its live leaves get consecutive rows behind those of the threads
// below it: the order of the list is a fixed function of the flags (deterministic), which is all a game's results need.
__global__ __launch_bounds__(kScanBlock) void tree_live_scan_kernel(Tree t, unsigned long long* __restrict__ count) {
    __shared__ int wave_total[kScanBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int rounds = (t.B + kScanBlock - 1) / kScanBlock;
    int cnt = 0;
    for (int r0 = 0; r0 < rounds; r0 += kScanUnroll) {
        int kind[kScanUnroll];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int g = (r0 + u) * kScanBlock + tid;
            kind[u] = (r0 + u < rounds && g < t.B) ? t.leaf_kind[g] : kLeafInactive;
        }
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) cnt += kind[u] == kLeafExpand ? 1 : 0;
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
    }
    if (lane == kWave - 1) wave_total[w] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int i = 0; i < kScanBlock / kWave; ++i) { if (i < w) before += wave_total[i]; total += wave_total[i]; }
    int row = before + incl - cnt;
    for (int r0 = 0; r0 < rounds; r0 += kScanUnroll) {
        int kind[kScanUnroll];
        Packed st[kScanUnroll];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const int g = (r0 + u) * kScanBlock + tid;
            const bool in = r0 + u < rounds && g < t.B;
            kind[u] = in ? t.leaf_kind[g] : kLeafInactive;          // second read: L2 hits
            if (in) st[u] = load_state(&t.leaf_state[g]);          // speculative: most leaves are live
        }
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            if (kind[u] != kLeafExpand) continue;
            const int g = (r0 + u) * kScanBlock + tid;
            t.live_row[g] = row;
            t.live_state[row] = st[u];
            ++row;
        }
    }
    if (tid == 0) *count = (unsigned long long)total;
}

// expand + backup of simulation s fused with the selection of simulation s+1 (same wave, same game: the edge
// records it just touched are still in L1/L2) -- one launch per simulation besides the network kernel.
// (forcing 8 waves / SIMD -- <= 96 SGPRs, 126 scalar spills -- was measured: no gain at 16 384 games, 1 % slower at C2)
template <bool IS_ROOT, bool COMPACT = false>
__global__ __launch_bounds__(kBlock) void tree_expand_select_kernel(Tree t, const float* __restrict__ lp1,
                                                                    const float* __restrict__ lp2,
                                                                    const float* __restrict__ lpm,
                                                                    const float* __restrict__ values,
                                                                    const float* __restrict__ noise, int noise_stride,
                                                                    float epsilon, int step) {
#ifndef LZ_EXP_NO_TREE_PRIO
    // The kernel is a chain of dependent loads with a few dozen instructions in between; in the two-stream search it
    // shares the SIMDs with the other half's network waves, which always have MFMAs to issue.  Raised wave priority
    // lets the short bursts between two loads go first.
    __builtin_amdgcn_s_setprio(3);
#endif
    LZ_EXPAND_SCRATCH(sc);
    const int g = wave_game();
    if (g >= t.B) return;
    const int lane = lane_id();
    RootInfo root;
    ptrdiff_t o = 0;
    if (COMPACT) o = (ptrdiff_t)(t.leaf_kind[g] == kLeafExpand ? t.live_row[g] : 0) - g;
#ifdef LZ_EXP_TREE_STAMPS
    const unsigned long long lz_t0 = __builtin_readcyclecounter();
#endif
    tree_expand<IS_ROOT>(t, g, lane, lp1 + o * 36, lp2 + o * 36, lpm + o * 36, nullptr, values + o, noise, noise_stride,
                         epsilon, sc, &root, step, nullptr, nullptr LZ_TSTAMP_PASS);
    __threadfence_block();
    if (IS_ROOT) root = load_root_info(t, g);                  // the root record itself was just written
    LZ_TSTAMP(g, 7)                                            // fence (+ root reload)
    tree_select(t, g, lane, root, -1, -1, nullptr LZ_TSTAMP_PASS);
#ifdef LZ_EXP_TREE_STAMPS
    LZ_TADD(g, 19, 1)
#endif
}

// The same step on TWO waves per game (round 6).  One wave's step is a chain of ~32 k cycles at the C2 launch shape (2 048
// games: two waves per SIMD, the kernel's duration IS that chain -- profiles/r05_pmc_sq_tree.md): 14 k of expansion, 2 k
// of backup, 16 k of descent.  The descent of simulation s + 1 does not depend on the expansion of simulation s except
// for ONE edge (the one the new node hangs on), so the game's first wave expands while its second backs up and descends;
// the critical path is the longer of the two (~20 k).  Hand-offs through an LDS word per game (tree_select, split_wait):
//   first wave : inputs in registers -> flag = 1;  node + edges written, release -> flag = 2
//   second wave: backup, fence, wait for 1 (it will overwrite the leaf record), descent (waits for 2 only if it reaches
//                the edge being expanded)
// Same stores in the same per-address order as the one-wave step: bit-identical trees (every tree parity test runs through
// this kernel; LZ_TREE_SPLIT=0 takes the one-wave kernel, tests/test_gpu_tree.py compares the two).  Used for launches of
// at most kSplitMaxGames games: at 16 384 games the SIMDs are issue-bound and twice the waves buy nothing.
constexpr int kSplitMaxGames = 8192;
template <bool COMPACT>
__global__ __launch_bounds__(kBlock) void tree_expand_select_split_kernel(Tree t, const float* __restrict__ lp1,
                                                                          const float* __restrict__ lp2,
                                                                          const float* __restrict__ lpm,
                                                                          const float* __restrict__ values, int step) {
    __builtin_amdgcn_s_setprio(3);
    LZ_EXPAND_SCRATCH(sc);
    __shared__ int s_flag[kWavesPerBlock / 2];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pair = wv >> 1, role = wv & 1;                      // role 0: expansion, role 1: backup + descent
    const int g = blockIdx.x * (kWavesPerBlock / 2) + pair;
    const int lane = lane_id();
    if (role == 0 && lane == 0) s_flag[pair] = 0;
    __syncthreads();                                               // the only workgroup barrier: before any early exit
    if (g >= t.B) return;
    volatile int* flag = &s_flag[pair];
    ptrdiff_t o = 0;
    if (COMPACT) o = (ptrdiff_t)(t.leaf_kind[g] == kLeafExpand ? t.live_row[g] : 0) - g;
    if (role == 0) {
        tree_expand<false, 1>(t, g, lane, lp1 + o * 36, lp2 + o * 36, lpm + o * 36, nullptr, values + o, nullptr, 0, 0.f, sc,
                              nullptr, step, flag);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // node, edges, the parent edge's child fields
        if (lane == 0) *flag = kSplitExpanded;
    } else {
        RootInfo root;
        SplitInfo split;
        tree_expand<false, 2>(t, g, lane, lp1 + o * 36, lp2 + o * 36, lpm + o * 36, nullptr, values + o, nullptr, 0, 0.f, sc,
                              &root, step, flag, &split);
        __threadfence_block();
        split_wait(flag, kSplitInputsRead);                        // the descent's end overwrites the leaf record
        tree_select(t, g, lane, root, split.wait_edge, split.nolegal_edge, flag);
    }
}

// ---- wave-batched leaves: the legacy search of src/mcts.py (batch_K leaves per tree and wave, no virtual loss) ------
// src/mcts.py:318-497.  Within a wave the statistics do not change, so the K leaves the reference collects with its
// restart-and-ban walks (:341-420) are simply the first K leaves of a depth-first traversal that visits the children
// of every node in descending PUCT order (first maximum first) -- the walk below continues that traversal instead of
// restarting it: per level it remembers the (score, index) of the last child it consumed and picks the next one
// in that order; a reserved / terminal / unexpanded child is a leaf, an expanded child is entered, a node without
// candidates is left.  Leaves go to slot j of the [batch_k][B] slot-major arrays; the slots' leaf states are the next
// network batch.  (Of the reference's safety limits the 128 back-steps per walk are modelled, see `ups` below; the 8K walks
// per wave only end futile repetition.  kWaveDepth levels cover every descent: a game lasts at most 144 plies,
// game_state.py:87-89 -- round 6; 48 before.)
constexpr int kWaveDepth = 160;
struct WaveArrays {
    int K, cap;
    int* path; int* path_len; int* leaf_kind; Packed* leaf_state; float* leaf_value; int* leaf_edge; int* leaf_parent;
    int* sims_done; int* unfinished;
    int* eval_row; Packed* eval_state; unsigned long long* eval_count;   // compact list of the leaves to evaluate
    unsigned long long* eval_total;                                      // running sum of eval_count (statistics)
    int max_back;                                                        // MAX_BACKTRACK_STEPS of the reference (128)
};
struct Level { int e0, ne_pl, parent_n, node; double last_sc; int last_k, in_edge; };
static_assert(sizeof(Level) == 32, "level record is 32 bytes");

__device__ __forceinline__ Tree slot_view(const Tree& t, const WaveArrays& w, int j) {
    Tree v = t;
    const size_t o = (size_t)j * t.B;
    v.path = w.path + o * w.cap; v.path_cap = w.cap; v.path_len = w.path_len + o; v.leaf_kind = w.leaf_kind + o;
    v.leaf_state = w.leaf_state + o; v.leaf_value = w.leaf_value + o; v.leaf_edge = w.leaf_edge + o;
    v.leaf_parent = w.leaf_parent + o;
    return v;
}

__device__ __forceinline__ void tree_select_wave(const Tree& t, const WaveArrays& w, int g, int lane,
                                                 const RootInfo& root, int sims, Level* stack, int* leaf_node,
                                                 int* leaf_act) {
    const int done = w.sims_done[g];
    int to_collect = sims - done;
    to_collect = to_collect < w.K ? to_collect : w.K;
    const bool live = t.root_terminal[g] == 0 && root.ne > 0 && to_collect > 0;
    const Node* nodes = t.nodes + (size_t)g * t.node_cap;
    const Edge* edges = t.edges;                               // pool indices
    int found = 0;
    if (live) {
        int d = 0;
        int e0 = root.e0, ne = root.ne, parent_n = root.visits, node_player = root.player, node = 0;
        double last_sc = INFINITY;
        int last_k = -1;
        // the current node's edge run and its scores stay in registers while the walk stays on this node
        Edge mine[2];
        double sc[2] = {0.0, 0.0};
        bool loaded = false;
        // The reference restarts every walk at the root and gives it up after MAX_BACKTRACK_STEPS = 128 upward moves
        // (src/mcts.py:337,371-391,404-414).  Within a wave nothing changes, so the walk for leaf j + 1 replays, move for
        // move, the traversal that found leaves 1..j before it goes on: its count is the traversal's CUMULATIVE number of
        // upward moves, and once it passes the limit this walk and -- being identical -- every further attempt of the
        // wave fails: the wave ends with the leaves found so far.  An upward move = leaving an exhausted internal node
        // for its parent, plus the step from a reserved leaf back to its parent when that parent has no other child left.
        int ups = 0;
        bool last_leaf = false;                                          // the current node's last consumed child was a leaf
        while (found < to_collect) {
            if (!loaded) {
                const double sq = sqrt((double)(parent_n > 1 ? parent_n : 1));
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = r * kWave + lane;
                    sc[r] = __builtin_nan("");                          // no child: never a candidate
                    if (k < ne) {
                        mine[r] = load_edge(&edges[(size_t)(e0 + k)]);
                        const int n = edge_n(mine[r].n_info);
                        double q = 0.0;
                        if (n > 0) {
                            const double mv = mine[r].W / (double)n;
                            const int child_player = (edge_info(mine[r].n_info) & kInfoWhite) ? -1 : 1;
                            q = child_player == node_player ? mv : -mv;
                        }
                        const double u = t.c_puct * (double)mine[r].P * sq / (1.0 + (double)n);
                        sc[r] = q + u;
                    }
                }
                loaded = true;
            }
            // next child of the current node in descending (score, -index) order
            double best = -INFINITY;
            int best_k = -1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = r * kWave + lane;
                const bool open = sc[r] < last_sc || (sc[r] == last_sc && k > last_k);    // not consumed yet
                if (open && sc[r] > best) { best = sc[r]; best_k = k; }
            }
            const double mx = lzw::wave_max(best);
            const uint64_t lo = __ballot(best_k >= 0 && best_k < kWave && best == mx);
            int chosen = -1;
            if (lo) chosen = __ffsll((unsigned long long)lo) - 1;
            else {
                const uint64_t hi = __ballot(best_k >= kWave && best == mx);
                if (hi) chosen = kWave + __ffsll((unsigned long long)hi) - 1;
            }
            if (chosen < 0) {                                           // nothing left below this node
                if (last_leaf && ++ups > w.max_back) break;             // reserved leaf -> this node (which has no alternative)
                if (d == 0) break;
                if (++ups > w.max_back) break;                          // this node -> its parent
                last_leaf = false;
                --d;
                const Level up = stack[d];
                e0 = up.e0; ne = up.ne_pl & 0xFF; node_player = (up.ne_pl & 0x100) ? -1 : 1; parent_n = up.parent_n;
                node = up.node; last_sc = up.last_sc; last_k = up.last_k;
                loaded = false;
                continue;
            }
            chosen = __builtin_amdgcn_readfirstlane(chosen);
            last_sc = mx; last_k = chosen;
            const int src = chosen & 63;
            const bool up2 = chosen >= kWave;
            const uint32_t c_ni = (uint32_t)lzw::lane_bcast((int)(up2 ? mine[1].n_info : mine[0].n_info), src);
            const int c_child = lzw::lane_bcast(up2 ? mine[1].child : mine[0].child, src);
            const int c_begin = lzw::lane_bcast(up2 ? mine[1].cbegin : mine[0].cbegin, src);
            const int c_meta = lzw::lane_bcast((int)(up2 ? mine[1].act : mine[0].act) | ((int)(up2 ? mine[1].cn : mine[0].cn) << 8), src);
            const uint8_t info = edge_info(c_ni);
            const int child_player = (info & kInfoWhite) ? -1 : 1;
            const int edge_id = e0 + chosen;
            const int entry = edge_id | (child_player != node_player ? (int)kPathFlip : 0);
            const bool terminal = (info & kInfoTerminal) != 0;
            if (terminal || c_child < 0) {                             // a leaf: slot `found`
                const size_t slot = (size_t)found * t.B + g;
                int* path = w.path + slot * w.cap;
                for (int l = lane; l < d; l += kWave) path[l] = stack[l + 1].in_edge;
                if (lane == 0) {
                    path[d] = entry;
                    w.path_len[slot] = d + 1;
                    w.leaf_kind[slot] = terminal ? kLeafTerminal : kLeafExpand;
                    w.leaf_value[slot] = terminal ? (float)((int)((info >> 2) & 3) - 1) : 0.f;
                    w.leaf_edge[slot] = edge_id;
                    w.leaf_parent[slot] = node;
                    leaf_node[found] = node;
                    leaf_act[found] = terminal ? -1 : (c_meta & 0xFF);
                }
                ++found;
                last_leaf = true;
                continue;
            }
            last_leaf = false;
            if (d + 1 >= kWaveDepth) continue;                          // deeper than a game can last: cannot happen
            if (lane == 0) {
                Level here;
                here.e0 = e0; here.ne_pl = ne | (node_player < 0 ? 0x100 : 0); here.parent_n = parent_n; here.node = node;
                here.last_sc = last_sc; here.last_k = last_k; here.in_edge = stack[d].in_edge;
                stack[d] = here;
                stack[d + 1].in_edge = entry;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            ++d;
            e0 = c_begin; ne = c_meta >> 8; parent_n = edge_n(c_ni); node_player = child_player; node = c_child;
            last_sc = INFINITY; last_k = -1;
            loaded = false;
        }
        // leaf positions for the network, one lane per leaf: parent position + action; the game's leaves take
        // consecutive rows of the compact evaluation list (one counter update per game)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int act = lane < found ? leaf_act[lane] : -1;
        const uint64_t need = __ballot(act >= 0);
        if (need) {
            unsigned long long base_row = 0ull;
            if (lane == 0) base_row = atomicAdd(w.eval_count, (unsigned long long)__popcll(need));
            base_row = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base_row >> 32)) << 32) |
                       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base_row & 0xFFFFFFFFull));
            if (act >= 0) {
                const int pn = leaf_node[lane];
                State leaf = unpack(pn == 0 ? root.state : load_state(&nodes[pn].state));
                int kd, p, q2, ex;
                index_to_code(leaf.phase, act, kd, p, q2, ex);
                apply_legal(leaf, kd, p, q2);
                const Packed ps = pack(leaf);
                const size_t slot = (size_t)lane * t.B + g;
                const unsigned long long row = base_row + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
                w.leaf_state[slot] = ps;
                w.eval_row[slot] = (int)row;
                w.eval_state[row] = ps;
            }
        }
    }
    for (int j = found + lane; j < w.K; j += kWave) w.leaf_kind[(size_t)j * t.B + g] = kLeafInactive;
    if (lane == 0) {
        int nd = done + found;
        if (live && found == 0) nd = sims;                              // nothing collectable: give the budget up
        w.sims_done[g] = nd;
        if (t.root_terminal[g] == 0 && root.ne > 0 && nd < sims) atomicAdd(w.unfinished, 1);
    }
}

__global__ __launch_bounds__(kBlock) void tree_select_wave_kernel(Tree t, WaveArrays w, int sims) {
    __shared__ Level s_stack[kWavesPerBlock][kWaveDepth];
    __shared__ int s_leaf_node[kWavesPerBlock][32], s_leaf_act[kWavesPerBlock][32];
    const int g = wave_game();
    if (g >= t.B) return;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __builtin_amdgcn_s_setprio(3);
    tree_select_wave(t, w, g, lane_id(), load_root_info(t, g), sims, s_stack[wv], s_leaf_node[wv], s_leaf_act[wv]);
}

__global__ __launch_bounds__(kBlock) void wave_budget_reset_kernel(WaveArrays w, int B, int reset_done) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g == 0) { *w.unfinished = 0; *w.eval_total += *w.eval_count; *w.eval_count = 0ull; }
    if (reset_done && g < B) w.sims_done[g] = 0;
}

// The wave's leaves in the reference's order (src/mcts.py:427-497): first every terminal leaf and every leaf whose
// position has no legal move is backed up (leaf order), then the evaluated leaves are expanded and backed up (leaf
// order).  Each step is the single-leaf expand of variant P on the slot's view of the per-leaf arrays; a workgroup
// fence (wait for the acknowledgements) between two steps keeps their updates of the same edges in order.
__global__ __launch_bounds__(kBlock) void tree_expand_wave_kernel(Tree t, WaveArrays w, const float* __restrict__ lp1,
                                                                  const float* __restrict__ lp2,
                                                                  const float* __restrict__ lpm,
                                                                  const float* __restrict__ priors220,
                                                                  const float* __restrict__ values, int slot_major) {
    LZ_EXPAND_SCRATCH(sc);
    const int g = wave_game();
    if (g >= t.B) return;
    const int lane = lane_id();
    __builtin_amdgcn_s_setprio(3);
    uint32_t second = 0;                                                 // slots handled by the second pass
    for (int pass = 0; pass < 2; ++pass) {
        for (int j = 0; j < w.K; ++j) {
            const size_t slot = (size_t)j * t.B + g;
            bool run;
            if (pass == 0) {
                const int kind = w.leaf_kind[slot];
                if (kind == kLeafInactive) break;                        // slots are filled in order
                run = kind == kLeafTerminal;
                if (kind == kLeafExpand) {
                    const State s = unpack(load_state(&w.leaf_state[slot]));
                    if (legal_count(legal_actions(s, 0)) == 0) run = true;
                    else second |= 1u << j;
                }
            } else {
                run = (second >> j) & 1u;
            }
            if (!run) continue;
            // evaluator rows: slot-major [batch_k][B], or the rows of the compact list the network evaluated
            // (tree_expand indexes its inputs with g, so the base pointers are shifted by row - g)
            const ptrdiff_t o = slot_major ? (ptrdiff_t)j * t.B
                                                     : (ptrdiff_t)(w.leaf_kind[slot] == kLeafExpand ? w.eval_row[slot] : 0) - g;
            tree_expand<false>(slot_view(t, w, j), g, lane, lp1 ? lp1 + o * 36 : nullptr, lp2 ? lp2 + o * 36 : nullptr,
                               lpm ? lpm + o * 36 : nullptr, priors220 ? priors220 + o * 220 : nullptr, values + o,
                               nullptr, 0, 0.f, sc);
            __threadfence_block();       // the next leaf's loads / atomics come after this leaf's stores / atomics
        }
    }
}

// ---- advance (a21): promote the played child to root, keep its subtree ---------------------------------------
// src/mcts.py:577-592, portable_mcts.py:74-87, portable_mcts.cpp:739-769.  One wave per game, in place:
//   1. mark the subtree of the chosen child: node ids ascend in expansion order, so a node is kept iff its parent
//      is kept -- 64 nodes per step, marks as 64-bit ballots in LDS, same-chunk chains resolved by iterating;
//   2. slide the kept nodes down to their rank (new id <= old id, chunk loads complete before chunk stores) and give
//      every kept edge run its new place: the runs are packed, in node order, into the game's OWN chunks from the
//      first one on (greedy, a run never straddles a chunk).  The runs were laid down in this order by the same
//      greedy rule, and packing a subsequence never gets ahead of packing the whole sequence, so a run's new place
//      is never behind its old one in the game's chunk order: the copy is safe in place;
//   3. copy the runs node by node (all loads of a run before its stores), translating child ids by rank and child
//      edge offsets from the moved nodes; chunks behind the last used one go back to the pool.
// A game whose child was never expanded or was re-seated gives all its chunks back and starts a fresh tree from
// root_state, exactly like lz_tree_begin.  Edge room is pooled, so only the NODE arena bounds a kept subtree: one that
// would not leave `reserve_nodes` nodes for the next search (the reference's tree is unbounded) is PRUNED, not dropped:
// marking stops at the first 64-node chunk that does not fit, so the oldest part of the subtree -- a prefix in
// expansion order, which is closed under "parent of" and holds the root's and the upper levels' statistics --
// survives; edges whose child fell past the cut keep their visit count and value sum and point to no node again (the
// next visit expands that position afresh).  `pruned` counts pruned games, `dropped` the (only defensive) whole-subtree
// drops.  Dynamic LDS: per wave ceil(node_cap / 64) mark words + as many prefix counts + 3 x 64 ints of batch offsets.
constexpr int kMarkWordsMax = 1024;        // four waves per workgroup: node_cap <= 65 536 (48 KB of marks + prefix counts)
constexpr int kMarkWordsBig = 8192;        // ONE wave per workgroup (round 6): node_cap <= 524 288 (96 KB) -- the arenas of a
                                           // long single-game search (MCTSCore with tens of thousands of simulations)

template <int WPB>
__global__ __launch_bounds__(WPB * kWave) void tree_advance_kernel(Tree t, const int* __restrict__ played_action,
                                                              const uint8_t* __restrict__ reset, int reserve_nodes,
                                                              int mark_words, int* __restrict__ dropped,
                                                              int* __restrict__ pruned, long long* __restrict__ ticks) {
    // `ticks` (measurement aid, NULL in production): per game 100 MHz ticks at the start / after the marks / after the
    // nodes / after the edge runs, and the numbers of nodes before and kept (lz_debug_advance_ticks)
    extern __shared__ uint64_t s_adv[];
    const int lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = blockIdx.x * WPB + wv;
    if (g >= t.B) return;
    Node* nodes = t.nodes + (size_t)g * t.node_cap;
    Edge* edges = t.edges;                                      // pool indices
    uint64_t* mark = s_adv + (size_t)wv * mark_words;
    int* nprefix = reinterpret_cast<int*>(s_adv + (size_t)WPB * mark_words) + (size_t)wv * mark_words;
    const uint64_t lt = (1ull << lane) - 1ull;
    auto rank_of = [&](int id) { return nprefix[id >> 6] + __popcll(mark[id >> 6] & ((1ull << (id & 63)) - 1ull)); };

    const Packed rs = t.root_state[g];
    const bool act = t.active == nullptr || t.active[g] != 0;
    const int ne = nodes[0].nedges, e0 = nodes[0].edge_begin;
    const int played = played_action != nullptr ? played_action[g] : -1;
    int c = -1, new_n = 0;
    double new_w = 0.0;
    if (act && !(reset != nullptr && reset[g]) && !t.root_terminal[g] && ne > 0 && played >= 0) {
        int found = -1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = r * kWave + lane;
            const uint64_t hit = __ballot(k < ne && (int)edges[(size_t)(e0 + (k < ne ? k : 0))].act == played);
            if (found < 0 && hit) found = r * kWave + __ffsll((unsigned long long)hit) - 1;
        }
        if (found >= 0) {
            const Edge E = edges[(size_t)(e0 + found)];
            if (E.child > 0 && !(edge_info(E.n_info) & kInfoTerminal)) {
                const Packed cs = nodes[E.child].state;     // must be the state the host moved to
                if (cs.w0 == rs.w0 && cs.w1 == rs.w1 && cs.w2 == rs.w2 && cs.w3 == rs.w3) {
                    c = E.child; new_n = edge_n(E.n_info); new_w = E.W;
                }
            }
        }
    }
    const int nn = t.n_nodes[g];
    const int words = (nn + 63) >> 6;
    int kept_nodes = 0;
    if (ticks != nullptr && lane == 0) { ticks[(size_t)g * 8] = (long long)wall_clock64(); ticks[(size_t)g * 8 + 4] = nn; }
    if (c > 0 && words > mark_words) {                          // more nodes than the mark words cover: fresh root
        c = -1;                                                 // (cannot happen: mark_words covers node_cap)
        if (lane == 0 && dropped != nullptr) atomicAdd(dropped, 1);
    }
    if (c > 0) {
        // ---- pass 1: marks + per-word prefix counts ----
        const int node_budget = t.node_cap - reserve_nodes;
        bool cut = false;
        // (the kernel's time is its slowest wave's -- a game that keeps thousands of nodes -- and that wave is bound by
        //  the latency of dependent loads: the parents of kAhead words are fetched together)
        constexpr int kAhead = 4;
        for (int w0 = 0; w0 < words && !cut; w0 += kAhead) {
            int par[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int id = (w0 + u) * kWave + lane;
                par[u] = id < nn ? nodes[id].parent : -1;
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int w = w0 + u;
                if (w >= words || cut) break;
                const int id = w * kWave + lane;
                const bool valid = id < nn;
                const int parent = par[u];
                const bool cand = valid && id > c && parent >= c;   // descendants have larger ids than their ancestors
                bool kept = valid && id == c;
                if (cand && parent < w * kWave) kept = (mark[parent >> 6] >> (parent & 63)) & 1ull;
                uint64_t bal = __ballot(kept);
                for (;;) {                                           // parents inside this chunk
                    const bool nk = kept || (cand && parent >= w * kWave && ((bal >> (parent - w * kWave)) & 1ull));
                    const uint64_t nb = __ballot(nk);
                    kept = nk;
                    if (nb == bal) break;
                    bal = nb;
                }
                const int chunk_nodes = __popcll(bal);
                if (kept_nodes + chunk_nodes > node_budget) {
                    // no room for this chunk: the subtree is cut here (expansion order), the rest is forgotten
                    for (int r = w + lane; r < words; r += kWave) { mark[r] = 0ull; nprefix[r] = kept_nodes; }
                    cut = true;
                    break;
                }
                if (lane == 0) { mark[w] = bal; nprefix[w] = kept_nodes; }
                kept_nodes += chunk_nodes;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (kept_nodes == 0) {                                   // defensive: nothing of the subtree fits (not reachable
            c = -1;                                              // with arenas that hold one search plus its reserve)
            if (lane == 0 && dropped != nullptr) atomicAdd(dropped, 1);
        } else if (cut && lane == 0 && pruned != nullptr) {
            atomicAdd(pruned, 1);
        }
    }
    if (c <= 0) {
        release_chunks_wave(t, g, 0, lane);
        if (lane == 0) begin_game(t, g, /*release=*/false);
        return;
    }
    if (ticks != nullptr && lane == 0) { ticks[(size_t)g * 8 + 1] = (long long)wall_clock64(); ticks[(size_t)g * 8 + 5] = kept_nodes; }
    // ---- pass 2: nodes, and the new places of their edge runs ----
    const int* list = t.chunk_list + (size_t)g * t.chunk_cap;
    const int n_chunks = t.n_chunks[g];
    const int CH = t.chunk;
    int ci = 0, off = 0;                                         // chunk being filled (index into the list), its fill
    bool broken = false;                                         // the packing ran past the game's chunk list (wave-uniform)
    int list_reg = lane < n_chunks ? list[lane] : 0;             // list[64 * (ci / 64) + lane]: no load on a chunk change
    int cbase = lzw::lane_bcast(list_reg, 0) * CH;
    // (the records of two words are fetched together: a word's stores land below its own ids, never on the next word's)
    for (int w0 = 0; w0 < words; w0 += 2) {
        // a node record as three 16-byte parts: the packed state (2 parts), then {edge_begin, nedges, parent, old_begin}
        u32x4 na[2], nb2[2], nc2[2];
        uint64_t ms[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int w = w0 + u;
            ms[u] = w < words ? mark[w] : 0ull;
            na[u] = nb2[u] = nc2[u] = (u32x4){0u, 0u, 0u, 0u};
            if ((ms[u] >> lane) & 1ull) {
                const u32x4* q = reinterpret_cast<const u32x4*>(&nodes[w * kWave + lane]);
                na[u] = q[0]; nb2[u] = q[1]; nc2[u] = q[2];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int w = w0 + u;
            const uint64_t m = ms[u];
            if (!m) continue;
            const int id = w * kWave + lane;
            const bool kept = (m >> lane) & 1ull;
            u32x4 meta = nc2[u];
            const int cnt = kept ? ((int)meta.y > 0 ? (int)meta.y : 0) : 0;
            int begin = 0, start = 0;
            for (;;) {
                const int c2 = lane >= start ? cnt : 0;
                const int incl = lzw::wave_incl_scan(c2);
                const uint64_t over = __ballot(c2 > 0 && off + incl > CH);
                const int first = over ? __ffsll((unsigned long long)over) - 1 : kWave;   // first run that does not fit
                if (lane >= start && lane < first) begin = cbase + off + incl - c2;
                const int placed = first == kWave ? lzw::lane_bcast(incl, kWave - 1)
                                                  : (lzw::lane_bcast(incl, first) - lzw::lane_bcast(c2, first));
                off += placed;
                if (!over) break;
                if (ci + 1 >= n_chunks) broken = true;          // (cannot happen: see the header comment -- and below)
                ci = ci + 1 < n_chunks ? ci + 1 : ci;
                if ((ci & 63) == 0) list_reg = ci + lane < n_chunks ? list[ci + lane] : 0;
                cbase = lzw::lane_bcast(list_reg, ci & 63) * CH;
                off = 0;
                start = first;
            }
            if (kept) {
                meta.w = meta.x;                                 // old_begin = edge_begin
                meta.x = (uint32_t)begin;
                meta.z = id == c ? 0xFFFFFFFFu : (uint32_t)rank_of((int)meta.z);
                u32x4* q = reinterpret_cast<u32x4*>(&nodes[nprefix[w] + __popcll(m & lt)]);
                q[0] = na[u]; q[1] = nb2[u]; q[2] = meta;
            }
        }
    }
    __threadfence_block();
    if (ticks != nullptr && lane == 0) ticks[(size_t)g * 8 + 2] = (long long)wall_clock64();
    if (broken) {
        // The in-place packing is only safe while a run's new place is never behind its old one, which holds as long as
        // every run was laid out by the same greedy rule (header comment).  Should a future change of the allocation rule
        // break that, the kept subtree is DROPPED and counted instead of being packed over live records
        release_chunks_wave(t, g, 0, lane);
        if (lane == 0) {
            if (dropped != nullptr) atomicAdd(dropped, 1);
            begin_game(t, g, /*release=*/false);
        }
        return;
    }
    // ---- pass 3: edge runs, FLAT over the kept edges of 64 nodes at a time.  The kernel's time is its slowest wave's --
    // a game that keeps well over a thousand nodes (a run of forced moves keeps the whole tree) -- and that wave is bound
    // by memory round trips, not by bytes: so a round moves kFlat x 64 records (about 48 nodes' runs) with ALL their loads
    // in flight, then all child-link lookups in flight, then the stores.  Lane l of slice r owns flat record
    // f = f0 + 64 r + l of the batch: its node is found by a 6-step binary search over the batch's run offsets (LDS).
    // Safe in place: a run's new place ends no later than its old place does, and old places ascend with the node, so
    // nothing a round stores lies in what a later round loads ----
    constexpr int kFlat = 12;
    int* bat = reinterpret_cast<int*>(s_adv + (size_t)WPB * mark_words) + (size_t)WPB * mark_words +
               (size_t)wv * (3 * kWave);
    int* bP = bat; int* bOb = bat + kWave; int* bNb = bat + 2 * kWave;
    for (int base = 0; base < kept_nodes; base += kWave) {
        const int id = base + lane;
        int ob = 0, nb = 0, cnt = 0;
        if (id < kept_nodes) {
            const u32x4 meta = reinterpret_cast<const u32x4*>(&nodes[id])[2];
            nb = (int)meta.x; cnt = (int)meta.y > 0 ? (int)meta.y : 0; ob = (int)meta.w;
        }
        const int incl = lzw::wave_incl_scan(cnt);
        const int T = lzw::lane_bcast(incl, kWave - 1);          // records of this batch
        bP[lane] = incl - cnt; bOb[lane] = ob; bNb[lane] = nb;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int f0 = 0; f0 < T; f0 += kFlat * kWave) {
            u32x4 lo[kFlat], hi[kFlat];
            int dst[kFlat];
#pragma unroll
            for (int r = 0; r < kFlat; ++r) {
                const int f = f0 + r * kWave + lane;
                dst[r] = -1;
                lo[r] = (u32x4){0u, 0u, 0u, 0u}; hi[r] = (u32x4){0xFFFFFFFFu, 0u, 0u, 0u};
                if (f0 + r * kWave < T) {                        // wave-uniform: slices past the batch issue nothing
                    int j = 0;
#pragma unroll
                    for (int st = 32; st > 0; st >>= 1) j += bP[j + st] <= f ? st : 0;
                    if (f < T) {
                        const int k = f - bP[j];
                        dst[r] = bNb[j] + k;
                        const u32x4* q = reinterpret_cast<const u32x4*>(&edges[(size_t)(bOb[j] + k)]);
                        lo[r] = q[0]; hi[r] = q[1];
                    }
                }
            }
            // child links: every lookup is issued unconditionally (node 0 where there is no kept child)
            int ncs[kFlat], ebs[kFlat];
            bool kc[kFlat];
#pragma unroll
            for (int r = 0; r < kFlat; ++r) {
                const int ch = (int)hi[r].x;
                const int chc = ch >= 0 ? ch : 0;
                kc[r] = ch >= 0 && ((mark[chc >> 6] >> (chc & 63)) & 1ull);
                ncs[r] = kc[r] ? rank_of(chc) : 0;
            }
#pragma unroll
            for (int r = 0; r < kFlat; ++r)
                ebs[r] = (f0 + r * kWave < T) ? nodes[ncs[r]].edge_begin : 0;
#pragma unroll
            for (int r = 0; r < kFlat; ++r) {
                if (dst[r] < 0) continue;
                if ((int)hi[r].x >= 0) {
                    if (kc[r]) { hi[r].x = (uint32_t)ncs[r]; hi[r].y = (uint32_t)ebs[r]; }
                    else { hi[r].x = 0xFFFFFFFFu; hi[r].y = 0u; hi[r].z &= ~0xFF00u; }   // past the cut: child -1, cbegin 0, cn 0
                }
                u32x4* q = reinterpret_cast<u32x4*>(&edges[(size_t)dst[r]]);
                q[0] = lo[r]; q[1] = hi[r];
            }
        }
        __builtin_amdgcn_wave_barrier();                          // the batch arrays are rewritten by the next batch
    }
    if (ticks != nullptr && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ticks[(size_t)g * 8 + 3] = (long long)wall_clock64(); }
    release_chunks_wave(t, g, ci + 1, lane);
    if (lane == 0) {
        t.n_nodes[g] = kept_nodes;
        t.n_edges[g] = cbase + off;          // off == chunk: "no open chunk" (the next run takes a new one)
        t.root_visits[g] = new_n;
        t.root_W[g] = new_w;
        t.root_init_value[g] = 0.f;
        t.path_len[g] = 0;
        t.root_terminal[g] = 0;
        t.leaf_kind[g] = kLeafReusedRoot;
        t.leaf_state[g] = rs;          // evaluated with the rest of the batch, result unused
        t.leaf_value[g] = 0.f;
    }
}

// ---- finish: root policy / pick -----------------------------------------------------------------------------
// policy = softmax(log(score)/T) over children with score > 0 (T <= 1e-6: one-hot argmax), score = N or, for the
//          training target, N + beta * normalised prior                        (portable_mcts.py:150-205, :690-700)
// pick   = uniform over the legal children (opening plies, :709-712), inverse-CDF sample of the selection policy
//          with the given uniform, or N -> Q -> P -> index                       (portable_mcts.py:208-261)
__device__ __forceinline__ void score_policy(const float (&sc)[2], const bool (&ok)[2], float temp, int lane,
                                             float (&pol)[2]) {
    const float smax = lzw::wave_max(fmaxf(ok[0] ? sc[0] : -1.f, ok[1] ? sc[1] : -1.f));
    if (temp <= 1e-6f) {
        const uint64_t lo = __ballot(ok[0] && sc[0] == smax);
        int am;
        if (lo) am = __ffsll((unsigned long long)lo) - 1;
        else am = kWave + __ffsll((unsigned long long)__ballot(ok[1] && sc[1] == smax)) - 1;
        pol[0] = (lane == am) ? 1.f : 0.f;
        pol[1] = (kWave + lane == am) ? 1.f : 0.f;
    } else {
        const float tt = fmaxf(temp, 1e-6f);
        const float lg0 = (ok[0] && sc[0] > 0.f) ? logf(sc[0]) / tt : -INFINITY;
        const float lg1 = (ok[1] && sc[1] > 0.f) ? logf(sc[1]) / tt : -INFINITY;
        const float mx = lzw::wave_max(fmaxf(lg0, lg1));
        const float e0f = lg0 == -INFINITY ? 0.f : expf(lg0 - mx), e1f = lg1 == -INFINITY ? 0.f : expf(lg1 - mx);
        const float sum = lzw::wave_sum(e0f + e1f);
        pol[0] = e0f / sum; pol[1] = e1f / sum;
    }
}

__global__ __launch_bounds__(kBlock) void tree_finish_kernel(Tree t, const float* __restrict__ temps,
                                                             const float* __restrict__ target_temps,
                                                             float prior_pseudocount,
                                                             const uint8_t* __restrict__ force_uniform,
                                                             int sample_moves,
                                                             const float* __restrict__ uniforms,
                                                             float* __restrict__ policy_dense,
                                                             int* __restrict__ chosen_index,
                                                             int4* __restrict__ chosen_code,
                                                             uint8_t* __restrict__ chosen_valid,
                                                             uint8_t* __restrict__ terminal_out,
                                                             float* __restrict__ root_value,
                                                             int* __restrict__ child_count,
                                                             int* __restrict__ child_action,
                                                             int* __restrict__ child_visits,
                                                             float* __restrict__ child_prior, int out_cap) {
    const int lane = lane_id();
    const int g = wave_game();
    if (g >= t.B) return;
    float* prow = policy_dense + (size_t)g * 220;
    for (int j = lane; j < 220; j += kWave) prow[j] = 0.f;
    const Node* nodes = t.nodes + (size_t)g * t.node_cap;
    const Edge* edges = t.edges;                               // pool indices
    const int ne = nodes[0].nedges;
    const State root = unpack(nodes[0].state);
    const bool term = t.root_terminal[g] != 0 || ne <= 0;
    if (lane == 0) {
        terminal_out[g] = term ? 1 : 0;
        chosen_valid[g] = term ? 0 : 1;
        chosen_index[g] = -1;
        chosen_code[g] = make_int4(-1, -1, -1, -1);
        child_count[g] = term ? 0 : ne;
        const int rv = t.root_visits[g];
        root_value[g] = term ? (ne == 0 && game_status(root) == 0 ? -1.f : (float)terminal_value_for_mover(root))
                             : (rv > 0 ? (float)(t.root_W[g] / (double)rv) : t.root_init_value[g]);
    }
    if (term) return;
    const int e0 = nodes[0].edge_begin;
    const float temp = temps[g];
    float v[2], pr[2]; int act[2]; double q[2]; bool ok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = r * kWave + lane;
        ok[r] = k < ne;
        const Edge e = edges[(size_t)(e0 + (ok[r] ? k : 0))];
        const int n = ok[r] ? edge_n(e.n_info) : 0;
        v[r] = (float)n;
        pr[r] = ok[r] ? e.P : 0.f;
        act[r] = ok[r] ? (int)e.act : 0;
        q[r] = 0.0;
        if (ok[r] && n > 0) {
            const double mv = e.W / (double)n;
            q[r] = ((edge_info(e.n_info) & kInfoWhite) ? -1 : 1) == root.player ? mv : -mv;
        }
        if (ok[r] && k < out_cap) {
            child_action[(size_t)g * out_cap + k] = act[r];
            child_visits[(size_t)g * out_cap + k] = n;
            child_prior[(size_t)g * out_cap + k] = pr[r];
        }
    }
    // ---- selection policy (drives the move), and the training target written to policy_dense ----
    float pol[2] = {0.f, 0.f};
    const float vmax = lzw::wave_max(fmaxf(ok[0] ? v[0] : -1.f, ok[1] ? v[1] : -1.f));
    score_policy(v, ok, temp, lane, pol);
    if (target_temps == nullptr && !(prior_pseudocount > 0.f)) {
#pragma unroll
        for (int r = 0; r < 2; ++r) if (ok[r]) prow[act[r]] = pol[r];
    } else {
        float sc[2] = {v[0], v[1]};
        if (prior_pseudocount > 0.f) {
            const float c0 = ok[0] ? fmaxf(pr[0], 1e-8f) : 0.f, c1 = ok[1] ? fmaxf(pr[1], 1e-8f) : 0.f;
            const float psum = lzw::wave_sum(c0 + c1);
            const bool bad = !(psum > 0.f) || !isfinite(psum);
            sc[0] = v[0] + prior_pseudocount * (bad ? 1.0f / (float)ne : c0 / psum);
            sc[1] = v[1] + prior_pseudocount * (bad ? 1.0f / (float)ne : c1 / psum);
        }
        float tpol[2] = {0.f, 0.f};
        score_policy(sc, ok, target_temps != nullptr ? target_temps[g] : temp, lane, tpol);
#pragma unroll
        for (int r = 0; r < 2; ++r) if (ok[r]) prow[act[r]] = tpol[r];
    }
    // ---- pick ----
    int pick = -1;
    if (force_uniform != nullptr && force_uniform[g] && uniforms != nullptr) {
        pick = (int)(uniforms[g] * (float)ne);
        pick = pick < 0 ? 0 : (pick >= ne ? ne - 1 : pick);
    } else if (sample_moves && uniforms != nullptr) {
        const float target = uniforms[g];
        float run = 0.f;
        int last = -1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float incl = pol[r];
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) { const float tt = __shfl_up(incl, o); if (lane >= o) incl += tt; }
            const uint64_t hb = __ballot(ok[r] && pol[r] > 0.f && run + incl > target);
            if (pick < 0 && hb) pick = r * kWave + __ffsll((unsigned long long)hb) - 1;
            const uint64_t vb = __ballot(ok[r] && pol[r] > 0.f);
            if (vb) last = r * kWave + 63 - __clzll((unsigned long long)vb);
            run += __shfl(incl, kWave - 1);
        }
        if (pick < 0) pick = last;
    } else {
        // most visits, then Q (atol 1e-6), then prior (atol 1e-8), then lowest index
        bool c0 = ok[0] && v[0] == vmax, c1 = ok[1] && v[1] == vmax;
        float qf0 = (float)q[0], qf1 = (float)q[1];
        const float qmax = lzw::wave_max(fmaxf(c0 ? qf0 : -INFINITY, c1 ? qf1 : -INFINITY));
        c0 = c0 && fabsf(qf0 - qmax) <= 1e-6f; c1 = c1 && fabsf(qf1 - qmax) <= 1e-6f;
        const float pmax = lzw::wave_max(fmaxf(c0 ? pr[0] : -INFINITY, c1 ? pr[1] : -INFINITY));
        c0 = c0 && fabsf(pr[0] - pmax) <= 1e-8f; c1 = c1 && fabsf(pr[1] - pmax) <= 1e-8f;
        const uint64_t lo = __ballot(c0);
        if (lo) pick = __ffsll((unsigned long long)lo) - 1;
        else { const uint64_t hi = __ballot(c1); if (hi) pick = kWave + __ffsll((unsigned long long)hi) - 1; }
    }
    if (pick >= 0 && lane == (pick & 63)) {
        const int r = pick >> 6;
        const int a = r == 0 ? act[0] : act[1];
        int kd, p, q2, ex;
        index_to_code(root.phase, a, kd, p, q2, ex);
        chosen_index[g] = a;
        chosen_code[g] = make_int4(kd, p, q2, ex);
    }
}

// =================================================================================================================
// Fused root-PUCT search (variant R, v1/python/mcts_gpu.py:1249-1457) on packed states: the host op chain
// encode -> project -> root_pack -> noise -> batch_apply_moves -> ... with data-dependent shapes and two host syncs
// becomes two fixed-shape kernels around the network launches; rows are padded to 72 actions, children are appended
// to one device-counted list that the network kernel consumes without a host round trip.
// =================================================================================================================
constexpr int kRootCap = 72;
// butterfly sum in the order of the stand-alone operators (lz_ops.hip), so that the fused path reproduces
// project_policy_logits_fast + root_pack_rows bit for bit
__device__ __forceinline__ float bfly_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void zero_u64_kernel(unsigned long long* p) { if (threadIdx.x == 0) *p = 0ull; }

// one wave per root: legal set (tensor semantics, fast_legal_mask.cpp:134-418), masked softmax of the combined head
// logits (project_policy_logits_fast.cpp:16-164), left-packed rows with renormalised priors (module.cpp:247-363),
// optional Dirichlet mix (mcts_gpu.py:1329-1339), child states (fast_apply_moves semantics)
__global__ __launch_bounds__(kBlock) void root_prepare_kernel(
    const Packed* __restrict__ roots, int64_t B, const float* __restrict__ lp1, const float* __restrict__ lp2,
    const float* __restrict__ lpm, const float* __restrict__ noise, float epsilon, int64_t* __restrict__ legal_index,
    float* __restrict__ priors, int4* __restrict__ codes, uint8_t* __restrict__ valid, int32_t* __restrict__ counts,
    uint8_t* __restrict__ terminal, float* __restrict__ leaf, Packed* __restrict__ child_states,
    int32_t* __restrict__ child_ref, unsigned long long* __restrict__ n_children, int64_t child_capacity,
    int32_t* __restrict__ overflow) {
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= B) return;
    const State s = unpack(roots[g]);
    const Legal L = legal_actions(s, /*fallback_forced=*/1);
    const int n = legal_count(L);
    float h1 = 0.f, h2 = 0.f, hm = 0.f;
    if (lane < kCells) { h1 = lp1[g * 36 + lane]; h2 = lp2[g * 36 + lane]; hm = lpm[g * 36 + lane]; }
    float v[4]; bool lg[4]; int slot[4];
    float mx = -INFINITY;
    int base = 0;
    bool fin = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int a = it * kWave + lane;
        int from = 0, dest = 0, cell = 0;
        bool dest_ok = false;
        if (a >= 36 && a < 180) {
            from = (a - 36) >> 2;
            const int d = move_dest(from, (a - 36) & 3);
            dest_ok = d >= 0;
            dest = dest_ok ? d : 0;
        } else if (a >= 180 && a < 216) cell = a - 180;
        else if (a < 36) cell = a;
        const float p1d = __shfl(h1, dest), p2f = __shfl(h2, from), p1c = __shfl(h1, cell), pmc = __shfl(hm, cell);
        const float x = a < 36 ? p1c : a < 180 ? (dest_ok ? p2f + p1d : -INFINITY) : a < 216 ? pmc : 0.f;
        lg[it] = a < 220 && legal_bit(L, a);
        v[it] = lg[it] ? x : -INFINITY;
        if (v[it] > mx) mx = v[it];
        fin = fin || isfinite(v[it]);
        const uint64_t bal = __ballot(lg[it]);
        slot[it] = base + __popcll(bal & ((1ull << lane) - 1ull));
        base += __popcll(bal);
    }
    const bool do_softmax = n > 0 && __ballot(fin) != 0ull;
    mx = lzw::wave_max(mx);
    float e[4], sum = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) { e[it] = (v[it] == -INFINITY) ? 0.f : expf(v[it] - mx); sum += e[it]; }
    sum = bfly_sum(sum);
    float pr[4], part = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) { pr[it] = do_softmax ? e[it] / sum : 0.f; if (lg[it]) part += pr[it]; }
    const float denom = fmaxf(bfly_sum(part), 1e-8f);
    float nz[4], nsum = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        pr[it] = pr[it] / denom;
        nz[it] = (noise != nullptr && lg[it] && slot[it] < kRootCap) ? noise[g * kRootCap + slot[it]] : 0.f;
        nsum += nz[it];
    }
    if (noise != nullptr && n > 1) {
        nsum = fmaxf(bfly_sum(nsum), 1e-8f);
        const float keep = 1.0f - epsilon;
#pragma unroll
        for (int it = 0; it < 4; ++it) pr[it] = keep * pr[it] + epsilon * (nz[it] / nsum);
    }
    // row padding, then the packed entries
    for (int j = lane; j < kRootCap; j += kWave) {
        legal_index[g * kRootCap + j] = -1;
        priors[g * kRootCap + j] = 0.f;
        codes[g * kRootCap + j] = make_int4(0, 0, 0, 0);
        valid[g * kRootCap + j] = 0;
        leaf[g * kRootCap + j] = 0.f;
    }
    unsigned long long cbase = 0;
    if (lane == 0) {
        counts[g] = n;
        terminal[g] = n == 0 ? 1 : 0;
        if (n > 0) cbase = atomicAdd(n_children, (unsigned long long)n);
    }
    cbase = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(cbase >> 32)) << 32) |
            (unsigned int)__builtin_amdgcn_readfirstlane((int)(cbase & 0xFFFFFFFFull));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // padding stores precede the entry stores of other lanes
    __builtin_amdgcn_wave_barrier();
    // the child list is sized for 72 children per root; a counter that was not reset (a caller bug) must not turn
    // into out-of-bounds stores: the row is dropped (no valid action) and the overflow is reported
    if ((int64_t)cbase + n > child_capacity) {
        if (lane == 0 && overflow != nullptr) atomicAdd(overflow, 1);
        return;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (!lg[it] || slot[it] >= kRootCap) continue;
        const int a = it * kWave + lane;
        int kd, p, q2, ex;
        index_to_code(s.phase, a, kd, p, q2, ex);
        const int64_t o = g * kRootCap + slot[it];
        legal_index[o] = a;
        priors[o] = pr[it];
        codes[o] = make_int4(kd, p, q2, ex);
        valid[o] = 1;
        State c = s;
        apply(c, kd, p, q2);
        child_states[cbase + slot[it]] = pack(c);
        child_ref[cbase + slot[it]] = (int32_t)o;
    }
}

// one lane per child: value from the parent's perspective, terminal children replaced by the soft material value
// (mcts_gpu.py:1352-1380: _child_values_to_parent_perspective, _terminal_mask_from_next_state, _soft_tanh_from_board_black)
__global__ __launch_bounds__(kBlock) void root_collect_kernel(const Packed* __restrict__ roots,
                                                              const Packed* __restrict__ child_states,
                                                              const int32_t* __restrict__ child_ref,
                                                              const float* __restrict__ child_values,
                                                              const unsigned long long* __restrict__ n_children,
                                                              int64_t capacity, float soft_k, float* __restrict__ leaf) {
    const int64_t n = (int64_t)(*n_children < (unsigned long long)capacity ? *n_children : (unsigned long long)capacity);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int32_t ref = child_ref[i];
        const State c = unpack(child_states[i]);
        const int parent_player = ((roots[ref / kRootCap].w0 >> 53) & 1) ? -1 : 1;
        float val = child_values[i];
        if (c.player != parent_player) val = -val;
        if (game_status(c) != 0) {
            const float delta = (float)(popc(c.black) - popc(c.white)) / 18.0f;
            val = tanhf(delta * soft_k) * (parent_player >= 0 ? 1.0f : -1.0f);
        }
        leaf[ref] = val;
    }
}

// ---- top-K lookahead of the root search (sparse_ply > 1; v1/python/mcts_gpu.py:976-1046, :1150-1160) ----------------
// root_topk_kernel: one wave per root.  The K best VALID children of the root by their current leaf value (highest first,
// lowest slot among equals) -> top_slot[g][k] (-1: the root has fewer than k + 1 legal actions) and the L2 position
// reached by that action (an all-zero record, phase 0 = no legal action, for the empty picks: lz_root_prepare then
// appends no children for it).  Rows of 72 slots: lane l holds slots l and l + 64.
__global__ __launch_bounds__(kBlock) void root_topk_kernel(const Packed* __restrict__ roots, int64_t B,
                                                           const float* __restrict__ leaf, const uint8_t* __restrict__ valid,
                                                           const int4* __restrict__ codes, int K,
                                                           int32_t* __restrict__ top_slot, Packed* __restrict__ l2_states) {
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= B) return;
    const int64_t row = g * kRootCap;
    const bool has1 = lane + kWave < kRootCap;
    float v0 = valid[row + lane] ? leaf[row + lane] : -INFINITY;
    float v1 = (has1 && valid[row + kWave + lane]) ? leaf[row + kWave + lane] : -INFINITY;
    const State s = unpack(roots[g]);
    for (int k = 0; k < K; ++k) {
        const float m = lzw::wave_max(fmaxf(v0, v1));
        int slot = -1;
        if (m > -INFINITY) {
            const uint64_t lo = __ballot(v0 == m);
            if (lo) slot = __ffsll((unsigned long long)lo) - 1;
            else { const uint64_t hi = __ballot(v1 == m); if (hi) slot = kWave + __ffsll((unsigned long long)hi) - 1; }
        }
        slot = __builtin_amdgcn_readfirstlane(slot);
        Packed out; out.w0 = out.w1 = out.w2 = out.w3 = 0ull;
        if (slot >= 0) {
            if (lane == (slot & 63)) {
                const int4 c = codes[row + slot];
                State child = s;
                apply(child, c.x, c.y, c.z);
                l2_states[g * K + k] = pack(child);
                if (slot < kWave) v0 = -INFINITY; else v1 = -INFINITY;
            }
        } else if (lane == 0) {
            l2_states[g * K + k] = out;
        }
        if (lane == 0) top_slot[g * K + k] = slot;
    }
}

// root_refine_kernel: one wave per root.  For every picked child: best value among ITS children as its own mover sees
// them (row g*K + k of the L2 leaf matrix; 0 when it has none or the maximum is not finite), and
// leaf[g][slot] = max(leaf[g][slot], that) -- the reference's refinement, kept for the picked slots only.
__global__ __launch_bounds__(kBlock) void root_refine_kernel(int64_t B, int K, const int32_t* __restrict__ top_slot,
                                                             const float* __restrict__ l2_leaf,
                                                             const uint8_t* __restrict__ l2_valid, float* __restrict__ leaf) {
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= B) return;
    const bool has1 = lane + kWave < kRootCap;
    for (int k = 0; k < K; ++k) {
        const int slot = top_slot[g * K + k];
        if (slot < 0) continue;                                    // wave-uniform
        const int64_t r2 = (g * K + k) * kRootCap;
        const float a = l2_valid[r2 + lane] ? l2_leaf[r2 + lane] : -INFINITY;
        const float b = (has1 && l2_valid[r2 + kWave + lane]) ? l2_leaf[r2 + kWave + lane] : -INFINITY;
        float best = lzw::wave_max(fmaxf(a, b));
        if (!isfinite(best)) best = 0.f;
        if (lane == 0) leaf[g * kRootCap + slot] = fmaxf(leaf[g * kRootCap + slot], best);
    }
}

// ---- per-game counter RNG (lz_rng.h): root noise Gammas and pick uniforms as pure functions of (seed, game, ply) ----
__global__ __launch_bounds__(kBlock) void rng_gamma_kernel(uint64_t seed, const int64_t* __restrict__ game,
                                                           const int64_t* __restrict__ ply, int64_t B, float alpha,
                                                           int count, float* __restrict__ out, int stride) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t g = i / count;
    const int k = (int)(i - g * count);
    if (g >= B) return;
    out[g * stride + k] = lzrng::gamma_draw(seed, game ? game[g] : g, ply ? ply[g] : 0, (uint32_t)k, alpha);
}
__global__ __launch_bounds__(kBlock) void rng_uniform_kernel(uint64_t seed, const int64_t* __restrict__ game,
                                                             const int64_t* __restrict__ ply, int64_t B, int purpose,
                                                             float* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= B) return;
    out[g] = lzrng::uniform_draw(seed, game ? game[g] : g, ply ? ply[g] : 0, (uint32_t)purpose);
}

WaveArrays make_wave(const LzTreeWaveDesc* w) {
    WaveArrays a;
    a.K = w->batch_k; a.cap = w->path_cap;
    a.path = w->path; a.path_len = w->path_len; a.leaf_kind = w->leaf_kind;
    a.leaf_state = reinterpret_cast<Packed*>(w->leaf_state); a.leaf_value = w->leaf_value; a.leaf_edge = w->leaf_edge;
    a.leaf_parent = w->leaf_parent; a.sims_done = w->sims_done; a.unfinished = w->unfinished;
    a.eval_row = w->eval_row; a.eval_state = reinterpret_cast<Packed*>(w->eval_state);
    a.eval_count = reinterpret_cast<unsigned long long*>(w->eval_count);
    a.eval_total = reinterpret_cast<unsigned long long*>(w->eval_total);
    a.max_back = w->max_backtrack_steps > 0 ? w->max_backtrack_steps : 128;
    return a;
}
bool wave_ok(const LzTreeWaveDesc* w) {
    return w && w->batch_k >= 1 && w->batch_k <= 32 && w->path_cap > kWaveDepth && w->path && w->path_len &&
           w->leaf_kind && w->leaf_state && w->leaf_value && w->leaf_edge && w->leaf_parent && w->sims_done &&
           w->unfinished && w->eval_row && w->eval_state && w->eval_count && w->eval_total;
}
inline unsigned gw(int64_t n) { return (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock); }
inline unsigned gw2(int64_t n) { return (unsigned)((n + kWavesPerBlock / 2 - 1) / (kWavesPerBlock / 2)); }   // two waves per game
// the split step (two waves per game) for launches that leave the SIMDs room for twice the waves; LZ_TREE_SPLIT=0: never
inline bool split_step(int64_t games) {
    const char* e = getenv("LZ_TREE_SPLIT");                   // read per call: the choice is frozen into a captured graph
    const char* m = getenv("LZ_TREE_SPLIT_MAX");               // experiment: another upper limit than kSplitMaxGames
    const int64_t limit = (m && m[0]) ? atoll(m) : kSplitMaxGames;
    return !(e && e[0] == '0') && games <= limit;
}
inline unsigned gt(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

extern "C" {

int lz_pack_states(const LzStateSoA* s, int64_t B, void* packed_out, void* stream) {
    if (B < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!s || !s->board || !s->marks_black || !s->marks_white || !s->phase || !s->current_player ||
        !s->pending_marks_required || !s->pending_marks_remaining || !s->pending_captures_required ||
        !s->pending_captures_remaining || !s->forced_removals_done || !s->move_count || !s->moves_since_capture ||
        !packed_out)
        return LZ_ERR_ARG;
    hipLaunchKernelGGL(pack_states_kernel, dim3(gt(B)), dim3(kBlock), 0, as_stream(stream), *s, B,
                       reinterpret_cast<Packed*>(packed_out));
    return st();
}

int lz_packed_to_model_input(const void* packed, int64_t B, float* out, void* stream) {
    if (B < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!packed || !out) return LZ_ERR_ARG;
    hipLaunchKernelGGL(packed_planes_kernel, dim3(gw(B)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const Packed*>(packed), B, out);
    return st();
}

int lz_tree_begin(const LzTreeDesc* d, void* stream) {
    if (!tree_ok(d)) return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    hipLaunchKernelGGL(tree_begin_kernel, dim3(gt(d->num_games)), dim3(kBlock), 0, as_stream(stream), make_tree(d));
    return st();
}

int lz_tree_select(const LzTreeDesc* d, void* stream) {
    if (!tree_ok(d)) return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    hipLaunchKernelGGL(tree_select_kernel, dim3(gw(d->num_games)), dim3(kBlock), 0, as_stream(stream), make_tree(d));
    return st();
}

int lz_tree_expand(const LzTreeDesc* d, int is_root, const float* lp1, const float* lp2, const float* lpmc,
                   const float* priors220, const float* values, const float* noise, int64_t noise_stride,
                   float epsilon, void* stream) {
    if (!tree_ok(d) || !values) return LZ_ERR_ARG;
    if (!priors220 && (!lp1 || !lp2 || !lpmc)) return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    const Tree t = make_tree(d);
    if (is_root)
        hipLaunchKernelGGL(tree_expand_kernel<true>, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, lp1, lp2,
                           lpmc, priors220, values, noise, (int)noise_stride, epsilon, -1);
    else
        hipLaunchKernelGGL(tree_expand_kernel<false>, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, lp1, lp2,
                           lpmc, priors220, values, nullptr, 0, 0.f, -1);
    return st();
}

int lz_tree_finish(const LzTreeDesc* d, const float* temperatures, const float* target_temperatures,
                   float prior_pseudocount, const uint8_t* force_uniform, int sample_moves, const float* uniforms,
                   float* policy_dense,
                   int32_t* chosen_index, int32_t* chosen_code, uint8_t* chosen_valid, uint8_t* terminal_mask,
                   float* root_value, int32_t* child_count, int32_t* child_action, int32_t* child_visits,
                   float* child_prior, int64_t out_cap, void* stream) {
    if (!tree_ok(d) || !temperatures || !policy_dense || !chosen_index || !chosen_code || !chosen_valid ||
        !terminal_mask || !root_value || !child_count || !child_action || !child_visits || !child_prior || out_cap < 1)
        return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    if (!(prior_pseudocount >= 0.f) || ((force_uniform || sample_moves) && !uniforms)) return LZ_ERR_ARG;
    hipLaunchKernelGGL(tree_finish_kernel, dim3(gw(d->num_games)), dim3(kBlock), 0, as_stream(stream), make_tree(d),
                       temperatures, target_temperatures, prior_pseudocount, force_uniform, sample_moves, uniforms, policy_dense, chosen_index, reinterpret_cast<int4*>(chosen_code),
                       chosen_valid, terminal_mask, root_value, child_count, child_action, child_visits, child_prior,
                       (int)out_cap);
    return st();
}

static long long* g_advance_ticks = nullptr;
int lz_debug_advance_ticks(int64_t* ticks) { g_advance_ticks = reinterpret_cast<long long*>(ticks); return LZ_OK; }

int lz_prof_aux_begin(int kind, void* stream);                       // lz_net.hip (bench.py roofline.secondary)
int lz_prof_aux_end(int kind, void* stream, int64_t units);

int lz_tree_advance(const LzTreeDesc* d, const int32_t* played_action, const uint8_t* reset, int64_t next_sims,
                    int32_t* dropped, int32_t* pruned, void* stream) {
    if (!tree_ok(d) || next_sims < 0) return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    const int words = (d->node_cap + kWave - 1) / kWave;
    if (words > kMarkWordsBig) return LZ_ERR_UNSUPPORTED;             // LDS mark words: <= 524 288 nodes per game
    const int64_t rn = next_sims + 1;
    if (rn > d->node_cap) return LZ_ERR_ARG;
    // a game's chunk list must be able to hold the worst case of its node arena (72 children everywhere), so that the
    // node budget is the only thing that can cut a kept subtree
    if ((int64_t)d->chunk_cap * (d->edge_chunk - (kMaxChildren - 1)) < (int64_t)d->node_cap * kMaxChildren) return LZ_ERR_ARG;
    const bool big = words > kMarkWordsMax;                           // arenas beyond 65 536 nodes: one wave per workgroup
    const int wpb = big ? 1 : kWavesPerBlock;
    const size_t lds = (size_t)wpb * (words * (sizeof(uint64_t) + sizeof(int)) + 3 * kWave * sizeof(int));
    if (big) {
        static std::mutex mu;
        static bool configured[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LZ_ERR_LAUNCH;
        std::lock_guard<std::mutex> lk(mu);
        if (!configured[dev]) {
            const size_t most = (size_t)kMarkWordsBig * (sizeof(uint64_t) + sizeof(int)) + 3 * kWave * sizeof(int);
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(tree_advance_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)most) != hipSuccess) {
                (void)hipGetLastError();
                return LZ_ERR_LAUNCH;
            }
            configured[dev] = true;
        }
    }
    (void)lz_prof_aux_begin(1, stream);
    if (big)
        hipLaunchKernelGGL(tree_advance_kernel<1>, dim3((unsigned)d->num_games), dim3(kWave), lds, as_stream(stream), make_tree(d),
                           played_action, reset, (int)rn, words, dropped, pruned, g_advance_ticks);
    else
        hipLaunchKernelGGL(tree_advance_kernel<kWavesPerBlock>, dim3(gw(d->num_games)), dim3(kBlock), lds, as_stream(stream),
                           make_tree(d), played_action, reset, (int)rn, words, dropped, pruned, g_advance_ticks);
    (void)lz_prof_aux_end(1, stream, d->num_games);
    return st();
}

// Whole search of one move, enqueued from C++ (no Python in the simulation loop, hipGraph-capturable):
//   begin -> [net -> expand_root + select] -> (sims-1) x [net -> expand+backup + select] -> net -> expand+backup
// `continue_trees`: the roots were prepared by lz_tree_advance (kept subtrees or fresh roots), so no begin.
static int tree_search_impl(const LzTreeDesc* d, const LzNetDesc* net, int64_t sims, float* planes, float* lp1,
                            float* lp2, float* lpmc, float* values, const float* noise, int64_t noise_stride,
                            float epsilon, bool continue_trees, void* stream) {
    if (!tree_ok(d) || !net || sims < 0 || !lp1 || !lp2 || !lpmc || !values) return LZ_ERR_ARG;
    const int64_t B = d->num_games;
    if (B == 0) return LZ_OK;
    int rc = continue_trees ? LZ_OK : lz_tree_begin(d, stream);
    if (rc) return rc;
    (void)planes;   // the network kernel stages its input straight from the 32-byte packed leaf states
    const Tree t = make_tree(d);
    if (t.live_count != nullptr) {
        // compact evaluation lists: simulation s evaluates live_count[s] leaves (see LzTreeDesc.live_*)
        if (d->live_count_cap < sims + 2) return LZ_ERR_ARG;
        hipLaunchKernelGGL(tree_live_scan_kernel, dim3(1), dim3(kScanBlock), 0, as_stream(stream), t, t.live_count);   // the roots
        for (int64_t s = 0; s <= sims; ++s) {
            rc = lz_net_forward_packed_counted_f16(net, d->live_state, B, d->live_count + s, lp1, lp2, lpmc, nullptr, values,
                                                   stream);
            if (rc) return rc;
            if (s == sims) {
                if (s == 0)
                    hipLaunchKernelGGL((tree_expand_kernel<true, true>), dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t,
                                       lp1, lp2, lpmc, (const float*)nullptr, values, noise, (int)noise_stride, epsilon, (int)s);
                else
                    hipLaunchKernelGGL((tree_expand_kernel<false, true>), dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t,
                                       lp1, lp2, lpmc, (const float*)nullptr, values, (const float*)nullptr, 0, 0.f, (int)s);
            } else if (s == 0) {
                hipLaunchKernelGGL((tree_expand_select_kernel<true, true>), dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream),
                                   t, lp1, lp2, lpmc, values, noise, (int)noise_stride, epsilon, (int)s);
            } else {
                (void)lz_prof_aux_begin(0, stream);
                if (split_step(t.B))
                    hipLaunchKernelGGL((tree_expand_select_split_kernel<true>), dim3(gw2(t.B)), dim3(kBlock), 0,
                                       as_stream(stream), t, lp1, lp2, lpmc, values, (int)s);
                else
                    hipLaunchKernelGGL((tree_expand_select_kernel<false, true>), dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream),
                                       t, lp1, lp2, lpmc, values, nullptr, 0, 0.f, (int)s);
                (void)lz_prof_aux_end(0, stream, B);
            }
            if (s < sims)                                           // the leaves of simulation s + 1
                hipLaunchKernelGGL(tree_live_scan_kernel, dim3(1), dim3(kScanBlock), 0, as_stream(stream), t,
                                   t.live_count + (s + 1));
        }
        return st();
    }
    for (int64_t s = 0; s <= sims; ++s) {
        rc = lz_net_forward_packed_f16(net, d->leaf_state, B, lp1, lp2, lpmc, nullptr, values, stream);
        if (rc) return rc;
        if (s == sims) {   // last simulation: nothing left to select
            if (s == 0)
                hipLaunchKernelGGL(tree_expand_kernel<true>, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, lp1, lp2,
                                   lpmc, (const float*)nullptr, values, noise, (int)noise_stride, epsilon, (int)s);
            else
                hipLaunchKernelGGL(tree_expand_kernel<false>, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, lp1, lp2,
                                   lpmc, (const float*)nullptr, values, (const float*)nullptr, 0, 0.f, (int)s);
        } else if (s == 0) {
            hipLaunchKernelGGL(tree_expand_select_kernel<true>, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, lp1,
                               lp2, lpmc, values, noise, (int)noise_stride, epsilon, (int)s);
        } else {
            (void)lz_prof_aux_begin(0, stream);                      // no-ops unless lz_prof_enable(1) (never in a capture)
            if (split_step(t.B))
                hipLaunchKernelGGL((tree_expand_select_split_kernel<false>), dim3(gw2(t.B)), dim3(kBlock), 0, as_stream(stream),
                                   t, lp1, lp2, lpmc, values, (int)s);
            else
                hipLaunchKernelGGL(tree_expand_select_kernel<false>, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, lp1,
                                   lp2, lpmc, values, nullptr, 0, 0.f, (int)s);
            (void)lz_prof_aux_end(0, stream, B);
        }
    }
    return st();
}

#ifdef LZ_EXP_TREE_STAMPS
LZ_API int lz_exp_tree_stamps(unsigned long long* out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tree_stamps), 32 * sizeof(unsigned long long)) != hipSuccess) return LZ_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tree_stamps), z, sizeof(z)) != hipSuccess) return LZ_ERR_LAUNCH;
    }
    return LZ_OK;
}
#endif

int lz_root_prepare(const void* root_states, int64_t B, const float* lp1, const float* lp2, const float* lpmc,
                    const float* noise, float epsilon, int64_t* legal_index_mat, float* priors_mat,
                    int32_t* action_code_mat, uint8_t* valid_mask, int32_t* counts, uint8_t* terminal_mask,
                    float* leaf_mat, void* child_states, int32_t* child_ref, uint64_t* n_children,
                    int64_t child_capacity, int32_t* overflow, void* stream) {
    if (B < 0) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!root_states || !lp1 || !lp2 || !lpmc || !legal_index_mat || !priors_mat || !action_code_mat || !valid_mask ||
        !counts || !terminal_mask || !leaf_mat || !child_states || !child_ref || !n_children || child_capacity < 0)
        return LZ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(action_code_mat) & 15) return LZ_ERR_ALIGN;
    // reset by a kernel, not hipMemsetAsync: a plain kernel node keeps stream order in every capture / replay mode
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long*>(n_children));
    hipLaunchKernelGGL(root_prepare_kernel, dim3(gw(B)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const Packed*>(root_states), B, lp1, lp2, lpmc, noise, epsilon, legal_index_mat,
                       priors_mat, reinterpret_cast<int4*>(action_code_mat), valid_mask, counts, terminal_mask, leaf_mat,
                       reinterpret_cast<Packed*>(child_states), child_ref,
                       reinterpret_cast<unsigned long long*>(n_children), child_capacity, overflow);
    return st();
}

int lz_root_collect(const void* root_states, const void* child_states, const int32_t* child_ref,
                    const float* child_values, const uint64_t* n_children, int64_t capacity, float soft_value_k,
                    float* leaf_mat, void* stream) {
    if (capacity < 0) return LZ_ERR_ARG;
    if (capacity == 0) return LZ_OK;
    if (!root_states || !child_states || !child_ref || !child_values || !n_children || !leaf_mat) return LZ_ERR_ARG;
    hipLaunchKernelGGL(root_collect_kernel, dim3(1024), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const Packed*>(root_states), reinterpret_cast<const Packed*>(child_states),
                       child_ref, child_values, reinterpret_cast<const unsigned long long*>(n_children), capacity,
                       soft_value_k, leaf_mat);
    return st();
}

int lz_root_topk_children(const void* root_states, int64_t B, const float* leaf_mat, const uint8_t* valid_mask,
                          const int32_t* action_code_mat, int64_t top_k, int32_t* top_slot, void* l2_states, void* stream) {
    if (B < 0 || top_k < 0 || top_k > kRootCap) return LZ_ERR_ARG;
    if (B == 0 || top_k == 0) return LZ_OK;
    if (!root_states || !leaf_mat || !valid_mask || !action_code_mat || !top_slot || !l2_states) return LZ_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(action_code_mat) & 15) return LZ_ERR_ALIGN;
    hipLaunchKernelGGL(root_topk_kernel, dim3(gw(B)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const Packed*>(root_states), B, leaf_mat, valid_mask,
                       reinterpret_cast<const int4*>(action_code_mat), (int)top_k, top_slot,
                       reinterpret_cast<Packed*>(l2_states));
    return st();
}

int lz_root_refine_topk(int64_t B, int64_t top_k, const int32_t* top_slot, const float* l2_leaf_mat,
                        const uint8_t* l2_valid_mask, float* leaf_mat, void* stream) {
    if (B < 0 || top_k < 0 || top_k > kRootCap) return LZ_ERR_ARG;
    if (B == 0 || top_k == 0) return LZ_OK;
    if (!top_slot || !l2_leaf_mat || !l2_valid_mask || !leaf_mat) return LZ_ERR_ARG;
    hipLaunchKernelGGL(root_refine_kernel, dim3(gw(B)), dim3(kBlock), 0, as_stream(stream), B, (int)top_k, top_slot,
                       l2_leaf_mat, l2_valid_mask, leaf_mat);
    return st();
}

int lz_tree_search(const LzTreeDesc* d, const LzNetDesc* net, int64_t sims, float* planes, float* lp1, float* lp2,
                   float* lpmc, float* values, const float* noise, int64_t noise_stride, float epsilon,
                   void* stream) {
    return tree_search_impl(d, net, sims, planes, lp1, lp2, lpmc, values, noise, noise_stride, epsilon, false, stream);
}

int lz_tree_search_continue(const LzTreeDesc* d, const LzNetDesc* net, int64_t sims, float* planes, float* lp1,
                            float* lp2, float* lpmc, float* values, const float* noise, int64_t noise_stride,
                            float epsilon, void* stream) {
    return tree_search_impl(d, net, sims, planes, lp1, lp2, lpmc, values, noise, noise_stride, epsilon, true, stream);
}

int lz_rng_gamma(uint64_t seed, const int64_t* game_id, const int64_t* ply, int64_t B, float alpha, int64_t count,
                 float* out, int64_t stride, void* stream) {
    if (B < 0 || count < 0 || count > 1024 || stride < count || !(alpha > 0.f)) return LZ_ERR_ARG;
    if (B == 0 || count == 0) return LZ_OK;
    if (!out) return LZ_ERR_ARG;
    hipLaunchKernelGGL(rng_gamma_kernel, dim3(gt(B * count)), dim3(kBlock), 0, as_stream(stream), seed, game_id, ply, B,
                       alpha, (int)count, out, (int)stride);
    return st();
}

int lz_rng_uniform(uint64_t seed, const int64_t* game_id, const int64_t* ply, int64_t B, int purpose, float* out,
                   void* stream) {
    if (B < 0 || purpose < 0 || purpose > 3) return LZ_ERR_ARG;
    if (B == 0) return LZ_OK;
    if (!out) return LZ_ERR_ARG;
    hipLaunchKernelGGL(rng_uniform_kernel, dim3(gt(B)), dim3(kBlock), 0, as_stream(stream), seed, game_id, ply, B, purpose,
                       out);
    return st();
}

int lz_tree_wave_select(const LzTreeDesc* d, const LzTreeWaveDesc* w, int64_t sims, int reset_budget, void* stream) {
    if (!tree_ok(d) || !wave_ok(w) || sims < 0) return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    const Tree t = make_tree(d);
    const WaveArrays a = make_wave(w);
    hipLaunchKernelGGL(wave_budget_reset_kernel, dim3(gt(t.B)), dim3(kBlock), 0, as_stream(stream), a, t.B, reset_budget);
    hipLaunchKernelGGL(tree_select_wave_kernel, dim3(gw(t.B)), dim3(kBlock), 0, as_stream(stream), t, a, (int)sims);
    return st();
}

int lz_tree_wave_expand(const LzTreeDesc* d, const LzTreeWaveDesc* w, const float* lp1, const float* lp2,
                        const float* lpmc, const float* priors220, const float* values, int slot_major, void* stream) {
    if (!tree_ok(d) || !wave_ok(w) || !values) return LZ_ERR_ARG;
    if (!priors220 && (!lp1 || !lp2 || !lpmc)) return LZ_ERR_ARG;
    if (d->num_games == 0) return LZ_OK;
    hipLaunchKernelGGL(tree_expand_wave_kernel, dim3(gw(d->num_games)), dim3(kBlock), 0, as_stream(stream), make_tree(d),
                       make_wave(w), lp1, lp2, lpmc, priors220, values, (slot_major || priors220) ? 1 : 0);
    return st();
}

// Whole wave-batched search of one move (hipGraph-capturable): begin -> net(roots) -> expand roots ->
// `waves` x [select batch_k leaves per game -> net(batch_k * B leaf slots) -> expand + backup in leaf order].
// lp1 / lp2 / lpmc / values hold batch_k * B rows.  Afterwards *wave->unfinished tells whether some game could not
// use its budget up (fewer open leaves than batch_k): the caller then runs further select / net / expand rounds.
int lz_tree_search_waves(const LzTreeDesc* d, const LzTreeWaveDesc* w, const LzNetDesc* net, int64_t sims, int64_t waves,
                         float* lp1, float* lp2, float* lpmc, float* values, const float* noise, int64_t noise_stride,
                         float epsilon, int continue_trees, int skip_roots, void* stream) {
    if (!tree_ok(d) || !wave_ok(w) || !net || sims < 0 || waves < 0 || !lp1 || !lp2 || !lpmc || !values) return LZ_ERR_ARG;
    const int64_t B = d->num_games;
    if (B == 0) return LZ_OK;
    int rc = LZ_OK;
    if (!skip_roots) {
        if (!continue_trees) { rc = lz_tree_begin(d, stream); if (rc) return rc; }
        rc = lz_net_forward_packed_f16(net, d->leaf_state, B, lp1, lp2, lpmc, nullptr, values, stream);
        if (rc) return rc;
        rc = lz_tree_expand(d, 1, lp1, lp2, lpmc, nullptr, values, noise, noise_stride, epsilon, stream);
        if (rc) return rc;
    }
    for (int64_t i = 0; i < waves; ++i) {
        rc = lz_tree_wave_select(d, w, sims, (!skip_roots && i == 0) ? 1 : 0, stream);
        if (rc) return rc;
        rc = lz_net_forward_packed_counted_f16(net, w->eval_state, (int64_t)w->batch_k * B, w->eval_count, lp1, lp2, lpmc,
                                               nullptr, values, stream);
        if (rc) return rc;
        rc = lz_tree_wave_expand(d, w, lp1, lp2, lpmc, nullptr, values, 0, stream);
        if (rc) return rc;
    }
    return st();
}

}  // extern "C"
