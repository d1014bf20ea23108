// lz_tree_dev.h -- device side of the device-resident full-tree PUCT search ("variant P"): record layouts and the
// per-game steps begin / select / expand + backup, each run by ONE wavefront for ONE game (see lz_engine.hip for the
// semantics and the reference lines they follow).  Included by lz_engine.hip (one kernel per step) and lz_search.hip (the
// persistent search kernel, where a workgroup runs these steps on the games it owns between two network passes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <math.h>
#include <stdint.h>

#include "lz_soa.h"
#include "lz_wave.h"

using namespace lz;

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kMaxChildren = 72;

// kLeafReusedRoot: the root survived an advance (a21) -- no evaluation needed, only the fresh noise mix
enum LeafKind : int { kLeafInactive = 0, kLeafExpand = 1, kLeafTerminal = 2, kLeafReusedRoot = 3 };
// edge info bits
constexpr uint8_t kInfoWhite = 1;       // child mover is white
constexpr uint8_t kInfoTerminal = 2;    // child is terminal (game over, or found to have no legal move)
// bits 2..3: terminal value + 1  (0 => -1, 1 => 0, 2 => +1), from the child's mover's perspective

// 32-byte edge record: one load brings everything the descent needs for a child, including where the child's
// own edges live -- select never touches node records until it has found the leaf's parent.
struct Edge {
    double W;            // value sum, child mover's perspective
    float P;             // prior
    uint32_t n_info;     // visit count (low 24 bits) | info (high 8 bits)
    int32_t child;       // node index or -1
    int32_t cbegin;      // child's first edge (valid when child >= 0)
    uint8_t act;         // 220-d action index
    uint8_t cn;          // child's edge count (valid when child >= 0)
    uint8_t pad[6];
};
static_assert(sizeof(Edge) == 32, "edge record is 32 bytes");
// 48-byte node record
struct Node {
    Packed state;
    int32_t edge_begin, nedges;      // first edge (index into the engine's edge pool), nedges = -1: not expanded
    int32_t parent;                  // parent node (-1 for the root)
    int32_t old_begin;               // scratch of lz_tree_advance: where the run lay before the compaction
};
static_assert(sizeof(Node) == 48, "node record is 48 bytes");
__device__ __forceinline__ int edge_n(uint32_t ni) { return (int)(ni & 0xFFFFFFu); }
__device__ __forceinline__ uint8_t edge_info(uint32_t ni) { return (uint8_t)(ni >> 24); }

// Edge storage (round 4): ONE pool per engine, cut into chunks of `chunk` edges (a power of two).  A game owns a list of
// chunks (chunk_list[g][0 .. n_chunks[g])), takes a new one from the free stack (free_chunks / pool_top) when the run
// of a new node does not fit into what is left of its open chunk -- runs never straddle chunks -- and gives chunks back
// when its tree is reset or compacted.  Every edge index in a record (Node::edge_begin, Edge::cbegin, path entries,
// leaf_edge) is an index into the pool, so the descent never looks at the chunk list.  Expand kernels only pop, begin /
// advance kernels only push: a pop never meets a push of the same launch.
struct Tree {
    int B, node_cap, chunk, path_cap, chunk_cap;
    const Packed* root_state;
    Node* nodes; Edge* edges;
    int* n_nodes; int* n_edges;      // n_edges[g]: next free pool index of the game's open chunk (multiple of chunk: none)
    int* chunk_list; int* n_chunks; int* free_chunks; int* pool_top;
    int* pool_stats;                 // [0] expansions refused because the pool was empty, [1] fewest free chunks seen,
                                     // [2] fresh roots that have not taken the chunk of their first expansion yet
    int* root_visits; double* root_W; float* root_init_value;
    int* path; int* path_len; int* leaf_kind; Packed* leaf_state; float* leaf_value;
    uint8_t* root_terminal; const uint8_t* active;
    int* leaf_edge; int* leaf_parent;      // edge / node the pending leaf hangs from (written by select)
    double c_puct;
    int fast_select;                 // tree_select: single-precision argmax where it provably equals the double one (LZ_TREE_F32SEL)
    // optional trace of what every expand step consumed (LzTreeDesc.trace_*; nullptr in production)
    int* trace_kind; Packed* trace_leaf; float* trace_heads; float* trace_priors; float* trace_value;
    int trace_cap;
    int* eval_count;      // optional per-game count of consumed evaluations
    // optional compact evaluation list (LzTreeDesc.live_*): the leaves that need the network, gathered after every select
    // step by an ordered one-workgroup scan (tree_live_scan_kernel), so that a network launch runs
    // ceil(live / samples-per-pass) passes instead of one per slot of the batch
    Packed* live_state; int* live_row; unsigned long long* live_count;
};

// Edge / node records are read with plain (L1 + L2 cached, normal retention) 16-byte loads.  This is safe next to the
// device-scope atomics of the backup because a launch never loads an edge line before its own atomics on it have
// completed: expand touches no edge record with a load, and a workgroup fence (= wait for the atomics' and stores'
// L2 acknowledgement) separates it from the selection that follows.  Streaming (`nt`) loads measured 10 % slower:
// they evict the upper tree levels from L2, which every simulation re-reads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Edge load_edge(const Edge* p) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    union U { u32x4 v[2]; Edge e; __device__ U() {} } u;
    u.v[0] = q[0]; u.v[1] = q[1];
    return u.e;
}
__device__ __forceinline__ Packed load_state(const Packed* p) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    union U { u32x4 v[2]; Packed s; __device__ U() {} } u;
    u.v[0] = q[0]; u.v[1] = q[1];
    return u.s;
}
constexpr uint32_t kPathFlip = 0x80000000u;    // path entry: edge index | flip bit (mover changes parent -> child)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_game() {
    return blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
#ifdef LZ_EXP_TREE_STAMPS   /* timing experiment (scripts/exp_tree_stamps.py): absolute clocks summed per stamp by ONE game's wave */
__device__ unsigned long long g_tree_stamps[32];
#define LZ_TSTAMP_ON(g) ((g) == 5)
#define LZ_TSTAMP(g, k) if (LZ_TSTAMP_ON(g)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        if (lane_id() == 0) { atomicAdd(&g_tree_stamps[k], (unsigned long long)__builtin_readcyclecounter() - lz_t0); \
                              atomicAdd(&g_tree_stamps[20 + (k)], 1ull); } }
#define LZ_TSTAMP_ARG , unsigned long long lz_t0 = 0
#define LZ_TSTAMP_PASS , lz_t0
#define LZ_TCLOCK(g, var) unsigned long long var = 0; if (LZ_TSTAMP_ON(g)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); var = __builtin_readcyclecounter(); }
#define LZ_TADD(g, k, v) if (LZ_TSTAMP_ON(g) && lane_id() == 0) atomicAdd(&g_tree_stamps[k], (unsigned long long)(v));
#else
#define LZ_TSTAMP(g, k)
#define LZ_TCLOCK(g, var)
#define LZ_TADD(g, k, v)
#define LZ_TSTAMP_ARG
#define LZ_TSTAMP_PASS
#endif

__device__ __forceinline__ double terminal_value_for_mover(const State& s) {   // portable_mcts.py:141-147
    const int st = game_status(s);
    if (st == 1 || st == -1) return st == s.player ? 1.0 : -1.0;
    return 0.0;
}


// ---- begin a search: fresh tree per game -----------------------------------------------------------------
// give chunks [keep, n_chunks) of game g back to the pool (one thread)
__device__ __forceinline__ void release_chunks(const Tree& t, int g, int keep) {
    const int nc = t.n_chunks[g];
    if (nc > keep) {
        const int base = atomicAdd(t.pool_top, nc - keep);
        const int* list = t.chunk_list + (size_t)g * t.chunk_cap;
        for (int i = keep; i < nc; ++i) t.free_chunks[base + (i - keep)] = list[i];
    }
    t.n_chunks[g] = keep < nc ? keep : nc;
}
// the same by a whole wave
__device__ __forceinline__ void release_chunks_wave(const Tree& t, int g, int keep, int lane) {
    const int nc = t.n_chunks[g];
    if (nc > keep) {
        int base = 0;
        if (lane == 0) base = atomicAdd(t.pool_top, nc - keep);
        base = __builtin_amdgcn_readfirstlane(base);
        const int* list = t.chunk_list + (size_t)g * t.chunk_cap;
        for (int i = keep + lane; i < nc; i += kWave) t.free_chunks[base + (i - keep)] = list[i];
        if (lane == 0) t.n_chunks[g] = keep;
    }
}
// Is game g's root a live fresh root that has not expanded yet?  pool_stats[2] counts exactly the games in this state
// (zero-filled arenas: n_nodes == 0, not pending), so every transition is booked as a difference: a second begin of a
// pending root, a begin after an advance that left a fresh root, a callback that failed before the root expansion or
// a refused root take leave the count where it is instead of leaking a chunk per event.
__device__ __forceinline__ bool root_pending(const Tree& t, int g) {
    return t.n_nodes[g] == 1 && t.nodes[(size_t)g * t.node_cap].nedges < 0 && t.root_terminal[g] == 0;
}

// one thread takes a chunk from the pool for game g: its first pool index, or -1 (pool empty / list full; counted).
// A chunk stays reserved for every fresh root that has not expanded yet (pool_stats[2] = the number of games whose root
// is in the root_pending state: begin kernels and the root's own take book the transitions): games progress
// independently, and a root whose expansion found the pool drained by the other games' deeper nodes would stay
// unexpanded for the whole search and end with an all-zero policy.
// Inside an expand launch the count only falls, so a stale read is conservative.
__device__ __forceinline__ int take_chunk(const Tree& t, int g, bool for_root = false) {
    const int nc = t.n_chunks[g];
    if (nc >= t.chunk_cap) { atomicAdd(t.pool_stats, 1); return -1; }
    const int top = atomicSub(t.pool_top, 1) - 1;            // free chunks left after this take
    const int reserved = for_root ? 0 : __atomic_load_n(t.pool_stats + 2, __ATOMIC_RELAXED);
    if (top < reserved) { atomicAdd(t.pool_top, 1); atomicAdd(t.pool_stats, 1); return -1; }
    if (for_root && root_pending(t, g)) atomicSub(t.pool_stats + 2, 1);
    atomicMin(t.pool_stats + 1, top);
    const int cid = t.free_chunks[top];
    t.chunk_list[(size_t)g * t.chunk_cap + nc] = cid;
    t.n_chunks[g] = nc + 1;
    return cid * t.chunk;
}

__device__ __forceinline__ void begin_game(const Tree& t, int g, bool release = true) {
    const bool was_pending = root_pending(t, g);
    if (release) release_chunks(t, g, 0);
    const Packed rs = t.root_state[g];
    Node& root = t.nodes[(size_t)g * t.node_cap];
    root.state = rs;
    root.edge_begin = 0;
    root.nedges = -1;                                      // unexpanded
    root.parent = -1;
    t.n_nodes[g] = 1;
    t.n_edges[g] = 0;
    t.root_visits[g] = 0;
    t.root_W[g] = 0.0;
    t.root_init_value[g] = 0.f;
    t.path_len[g] = 0;
    const State s = unpack(rs);
    const bool act = t.active == nullptr || t.active[g] != 0;
    const bool term = game_status(s) != 0;                // portable_mcts.py:601-603
    t.root_terminal[g] = (term || !act) ? 1 : 0;
    t.leaf_kind[g] = (term || !act) ? kLeafInactive : kLeafExpand;
    const bool pending = !(term || !act);                  // this root will need the chunk of its first expansion
    if (pending != was_pending) atomicAdd(t.pool_stats + 2, pending ? 1 : -1);
    t.leaf_state[g] = rs;                                  // the root is the first pending evaluation
    t.leaf_value[g] = 0.f;
}

// ---- select: one wave per game ---------------------------------------------------------------------------
// Per level the only dependent load is the current node's edge run (32 B per lane, coalesced); the reduction is
// DPP-based and the chosen edge is broadcast with v_readlane.  The chosen child's state record is fetched
// speculatively next to the next level's edge run, so reaching the leaf costs no extra round trip.
struct RootInfo { int ne, e0, visits, player; Packed state; };
__device__ __forceinline__ RootInfo load_root_info(const Tree& t, int g) {
    RootInfo r;
    const Node* root = t.nodes + (size_t)g * t.node_cap;
    r.state = load_state(&root->state);
    r.ne = root->nedges; r.e0 = root->edge_begin;
    r.visits = t.root_visits[g];
    r.player = ((r.state.w0 >> 53) & 1) ? -1 : 1;
    return r;
}

// Split step (tree_expand_select_split_kernel): the descent runs on the game's SECOND wave while the first one still
// expands the previous simulation's leaf.  Nothing the descent reads depends on that expansion except the child fields of
// ONE edge (`wait_edge`: the edge the new node hangs on) -- reaching it, the descent waits for the partner's flag (LDS,
// workgroup-scope release / acquire: both waves share the CU's L1) and re-reads the record; `nolegal_edge`: the previous
// leaf turned out to have no legal move, which the partner records as a terminal flag on that edge (value -1).
constexpr int kSplitInputsRead = 1, kSplitExpanded = 2;
__device__ __forceinline__ void split_wait(volatile int* flag, int phase) {
    while (*flag < phase) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void tree_select(const Tree& t, int g, int lane, const RootInfo& root, int wait_edge = -1,
                                            int nolegal_edge = -1, volatile int* flag = nullptr LZ_TSTAMP_ARG) {
    if (t.root_terminal[g]) { if (lane == 0) t.leaf_kind[g] = kLeafInactive; return; }
    const Node* nodes = t.nodes + (size_t)g * t.node_cap;
    const Edge* edges = t.edges;                               // pool indices
    int* path = t.path + (size_t)g * t.path_cap;
    int node = 0, depth = 0;
    int parent_n = root.visits;
    int node_player = root.player;
    int ne = root.ne, e0 = root.e0;
    Packed node_state = root.state;
    int kind = kLeafInactive;
    float term_value = 0.f;
    int leaf_action = 0, leaf_edge = -1;
    while (ne > 0) {
        LZ_TCLOCK(g, lv_t0)
        // the node's run: a wave-uniform base (scalar 64-bit arithmetic) + a 32-bit lane offset per load
        const Edge* run = edges + (size_t)__builtin_amdgcn_readfirstlane(e0);
        Edge mine[2];
        mine[1] = Edge{};
        // up to 2 children per lane, ascending edge index; the second slot only exists for nodes with more than 64
        // children (a wave-uniform branch: the common case runs half the score arithmetic)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && ne <= kWave) break;
            const int k = r * kWave + lane;
            if (k < ne) mine[r] = load_edge(&run[k]);
        }
#ifdef LZ_EXP_TREE_STAMPS
        if (LZ_TSTAMP_ON(g)) { LZ_TCLOCK(g, lv_t1) LZ_TADD(g, 16, lv_t1 - lv_t0) lv_t0 = lv_t1; }   // wait for the run
#endif
        int chosen = -1;
        // ---- the argmax in single precision, accepted only when it is PROVABLY the double-precision argmax (round 6; opt-in:
        // same-box A/B on the bench's random-init nets 19.0 - 19.7 -> 19.7 - 20.2 us at 2 048 games, 67.0 -> 68.9 us at 16 384,
        // profiles/r06_tree_f32sel_ab.jsonl -- their priors are nearly flat, so most levels end in the double path anyway) ----
        // The reference scores are doubles, and visit counts must come out bit for bit -- but the argmax of a level needs
        // the exact arithmetic only when its two best candidates are close.  With |W / n| <= 1 and u >= 0 a score s
        // satisfies |q| + |u| <= 2 + |s|, so five correctly rounded fp32 operations are off by at most 3e-7 (2 + |s|) <
        // 1e-6 (1 + |s|) from the double value: a candidate that leads every other one by more than 1e-4 (1 + |s|) in
        // fp32 (100 x that bound) leads in double too, and is the unique maximum (no tie to break).  Otherwise -- near
        // ties, exact ties (equal priors on unvisited children), NaN / non-finite scores -- the double path below decides
        // as before; so does a level on which some |W / n| exceeds 1 (an external evaluator with another value scale: the
        // bound above assumes values in [-1, 1]).  The double arithmetic of a level (sqrt, two divisions, a 64-bit wave
        // maximum) was the largest single item of the step (profiles/r05_pmc_sq_tree.md: 1.76 k of ~2.6 k cycles per level).
        if (t.fast_select) {
            const float sqf = sqrtf((float)(parent_n > 1 ? parent_n : 1));
            const float cf = (float)t.c_puct;
            float fs[2] = {-INFINITY, -INFINITY};
            bool off_scale = false;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && ne <= kWave) break;
                if (r * kWave + lane < ne) {
                    const int n = edge_n(mine[r].n_info);
                    float q = 0.f;
                    if (n > 0) {
                        const float mv = (float)mine[r].W / (float)n;
                        const int child_player = (edge_info(mine[r].n_info) & kInfoWhite) ? -1 : 1;
                        q = child_player == node_player ? mv : -mv;
                        off_scale = off_scale || !(fabsf(mv) <= 1.0001f);
                    }
                    const float s = q + cf * mine[r].P * sqf / (1.0f + (float)n);
                    fs[r] = s == s ? s : -INFINITY;             // a NaN is never a candidate (and sends the level to the double path
                }                                               // if nothing else is one)
            }
            const float b = fs[1] > fs[0] ? fs[1] : fs[0];
            const float m1 = lzw::wave_max_nonan(b);
            const uint64_t w0 = __ballot(fs[0] == m1), w1 = __ballot(fs[1] == m1);
            if (m1 > -INFINITY && m1 < INFINITY && __popcll(w0) + __popcll(w1) == 1 && __ballot(off_scale) == 0ull) {
                const int c = w0 ? __ffsll((unsigned long long)w0) - 1 : kWave + __ffsll((unsigned long long)w1) - 1;
                const float b2 = (c & 63) == lane ? (c < kWave ? fs[1] : fs[0]) : b;
                const float m2 = lzw::wave_max_nonan(b2);
                if (m1 - m2 > 1e-4f * (1.0f + fabsf(m1))) chosen = c;        // m2 = -inf (a single child): +inf
            }
        }
        if (chosen < 0) {
            const double sq = sqrt((double)(parent_n > 1 ? parent_n : 1));
            double best = -INFINITY;
            int best_k = -1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && ne <= kWave) break;
                const int k = r * kWave + lane;
                if (k < ne) {
                    const int n = edge_n(mine[r].n_info);
                    double q = 0.0;
                    if (n > 0) {
                        const double mv = mine[r].W / (double)n;
                        const int child_player = (edge_info(mine[r].n_info) & kInfoWhite) ? -1 : 1;
                        q = child_player == node_player ? mv : -mv;
                    }
                    const double u = t.c_puct * (double)mine[r].P * sq / (1.0 + (double)n);
                    const double sc = q + u;
                    if (sc > best) { best = sc; best_k = k; }
                }
            }
            const double mx = lzw::wave_max(best);
            const uint64_t lo = __ballot(best_k >= 0 && best_k < kWave && best == mx);   // lowest index among the maxima
            if (lo) chosen = __ffsll((unsigned long long)lo) - 1;
            else {
                const uint64_t hi = __ballot(best_k >= kWave && best == mx);
                if (!hi) break;                                    // every score NaN (portable: best_child is None)
                chosen = kWave + __ffsll((unsigned long long)hi) - 1;
            }
        }
        chosen = __builtin_amdgcn_readfirstlane(chosen);
        const int src = chosen & 63;
        const bool up = chosen >= kWave;
        const uint32_t c_ni = (uint32_t)lzw::lane_bcast((int)(up ? mine[1].n_info : mine[0].n_info), src);
        int c_child = lzw::lane_bcast(up ? mine[1].child : mine[0].child, src);
        int c_begin = lzw::lane_bcast(up ? mine[1].cbegin : mine[0].cbegin, src);
        int c_meta = lzw::lane_bcast((int)(up ? mine[1].act : mine[0].act) | ((int)(up ? mine[1].cn : mine[0].cn) << 8), src);
        uint8_t info = edge_info(c_ni);
        if (e0 + chosen == wait_edge) {                        // the partner wave is hanging a node on this edge right now
            split_wait(flag, kSplitExpanded);
            const Edge fresh = load_edge(&run[chosen]);
            c_child = fresh.child; c_begin = fresh.cbegin;
            c_meta = (int)fresh.act | ((int)fresh.cn << 8);
        }
        if (e0 + chosen == nolegal_edge) info |= kInfoTerminal;   // value bits 0 = -1 (tree_expand, n == 0)
        const int child_player = (info & kInfoWhite) ? -1 : 1;
#ifdef LZ_EXP_TREE_STAMPS
        { LZ_TCLOCK(g, lv_t2) LZ_TADD(g, 17, lv_t2 - lv_t0) LZ_TADD(g, 18, 1) }                                        // scores, maximum, broadcast
#endif
        leaf_edge = e0 + chosen;
        if (lane == 0) path[depth] = leaf_edge | (child_player != node_player ? (int)kPathFlip : 0);
        ++depth;
        if (info & kInfoTerminal) {
            kind = kLeafTerminal;
            term_value = (float)((int)((info >> 2) & 3) - 1);
            break;
        }
        if (c_child < 0) { kind = kLeafExpand; leaf_action = c_meta & 0xFF; break; }
        node_state = load_state(&nodes[c_child].state);     // in flight together with the next level's edges
        parent_n = edge_n(c_ni);
        node_player = child_player;
        node = c_child;
        e0 = c_begin;
        ne = c_meta >> 8;
        if (depth >= t.path_cap - 1) break;
    }
    LZ_TSTAMP(g, 8)                                            // descent done
    if (lane == 0) {
        t.path_len[g] = depth;
        t.leaf_kind[g] = kind;
        t.leaf_value[g] = term_value;
        t.leaf_edge[g] = leaf_edge;
        t.leaf_parent[g] = node;
        if (kind == kLeafExpand) {
            State leaf = unpack(node_state);
            int kd, p, q2, ex;
            index_to_code(leaf.phase, leaf_action, kd, p, q2, ex);
            apply_legal(leaf, kd, p, q2);                   // an action this engine enumerated: no re-validation
            t.leaf_state[g] = pack(leaf);
        }
    }
    LZ_TSTAMP(g, 9)                                            // leaf state written
}

// LDS scratch of one wave's expand step: the legal actions of the leaf, compacted (value, action index)
struct ExpandScratch { float* val; int* act; };          // 80 entries each
constexpr int kExpandScratchBytes = 80 * 8;
#define LZ_EXPAND_SCRATCH(name)                                                            \
    __shared__ float name##_val[kWavesPerBlock][80];                                      \
    __shared__ int name##_act[kWavesPerBlock][80];                                        \
    const int name##_wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));        \
    const ExpandScratch name = {name##_val[name##_wv], name##_act[name##_wv]}

// ---- expand (+ backup): one wave per game -----------------------------------------------------------------
// priors come either from the three 36-wide log-prob heads (production) or from a dense 220-d prior row
// (injected evaluator, parity runs).  IS_ROOT: no backup, optional noise mix.
// Every load that does not depend on another load is issued up front (leaf record, evaluator outputs, allocation
// counters, the path, the root's statistics); the backup is fire-and-forget device atomics (N += 1 on the count
// field, W += v in double -- one addition per edge and simulation, so bit-identical to a read-modify-write), the
// signs come from the flip bits select left in the path entries.  Dependent round trips: 1 (was 5).
// ROLE 0: the whole step on one wave.  Split step (two waves per game): ROLE 1 = the expansion only (signals `flag`: its inputs
// are in registers / its node and edges are written), ROLE 2 = the backup only (+ what the descent must know about the
// leaf, `split`: the edge a node is being hung on, or the edge of a leaf without a legal move).
struct SplitInfo { int wait_edge, nolegal_edge; };
template <bool IS_ROOT, int ROLE = 0>
__device__ __forceinline__ void tree_expand(const Tree& t, int g, int lane, const float* __restrict__ lp1,
                                            const float* __restrict__ lp2, const float* __restrict__ lpm,
                                            const float* __restrict__ priors220, const float* __restrict__ values,
                                            const float* __restrict__ noise, int noise_stride, float epsilon,
                                            ExpandScratch sc, RootInfo* root_after = nullptr, int step = -1,
                                            volatile int* flag = nullptr, SplitInfo* split = nullptr LZ_TSTAMP_ARG) {
    static_assert(ROLE == 0 || !IS_ROOT, "the root step is never split");
    LZ_TSTAMP(g, 0)
    const int kind = t.leaf_kind[g];
    Node* nodes = t.nodes + (size_t)g * t.node_cap;
    Edge* edges = t.edges;                                     // pool indices
    const int* path = t.path + (size_t)g * t.path_cap;
    // ---- independent loads, all in flight together ----
    RootInfo root{};
    int plen_ld = 0, leaf_edge = -1, leaf_parent = 0, nn_ld = 0, ne_ld = 0, path_entry = 0;
    double root_w = 0.0;
    float leaf_value_ld = 0.f, value_ld = 0.f;
    Packed leaf_packed = t.leaf_state[g];
    if (!IS_ROOT) {
        root = load_root_info(t, g);
        plen_ld = t.path_len[g];
        leaf_edge = t.leaf_edge[g];
        leaf_parent = t.leaf_parent[g];
        path_entry = lane < t.path_cap ? path[lane] : 0;       // first 64 entries
        root_w = t.root_W[g];
        leaf_value_ld = t.leaf_value[g];
    }
    nn_ld = t.n_nodes[g];
    ne_ld = t.n_edges[g];
    value_ld = values[g];
    if (root_after != nullptr) *root_after = root;
    if (ROLE == 1) {                                           // the partner may now overwrite the leaf record (next leaf)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *flag = kSplitInputsRead;
    }
    if (ROLE == 2) { split->wait_edge = -1; split->nolegal_edge = -1; }
    LZ_TSTAMP(g, 1)                                            // the independent loads have arrived
    // trace slot of this step (parity tests only; wave-uniform)
    const bool tracing = ROLE != 2 && t.trace_kind != nullptr && step >= 0 && step < t.trace_cap;
    const size_t tslot = tracing ? (size_t)step * (size_t)t.B + (size_t)g : 0;
    if (tracing) {
        if (lane == 0) { t.trace_kind[tslot] = kind; t.trace_leaf[tslot] = leaf_packed; t.trace_value[tslot] = value_ld; }
        for (int a = lane; a < 220; a += kWave) t.trace_priors[tslot * 220 + a] = 0.f;
    }
    if (kind == kLeafInactive) return;
    if (kind == kLeafReusedRoot) {
        // portable_mcts.py:302-317 / :617-621: a root kept by advance_root gets a fresh noise mix on its
        // existing priors, renormalised by max(sum, 1e-8); nothing else happens before the first selection.
        if (IS_ROOT && noise != nullptr) {
            const int ne = nodes[0].nedges, e0 = nodes[0].edge_begin;
            if (ne > 1) {
                const float keep = (float)(1.0 - (double)epsilon);
                float pr[2]; bool ok[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = r * kWave + lane;
                    ok[r] = k < ne;
                    pr[r] = ok[r] ? keep * edges[(size_t)(e0 + k)].P + epsilon * noise[(size_t)g * noise_stride + k] : 0.f;
                }
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    uint64_t m = __ballot(ok[r]);
                    while (m) {
                        const int l = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m) - 1);
                        psum += lzw::lane_bcast(pr[r], l);
                        m &= m - 1;
                    }
                }
                const float denom = psum < 1e-8f ? 1e-8f : psum;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (ok[r]) edges[(size_t)(e0 + r * kWave + lane)].P = pr[r] / denom;
            }
        }
        return;
    }
    const int plen = IS_ROOT ? 0 : plen_ld;
    double backup_value = 0.0;

    if (kind == kLeafTerminal) {
        backup_value = (double)leaf_value_ld;
    } else {
        if (ROLE != 2 && t.eval_count != nullptr && lane == 0) t.eval_count[g] += 1;     // this game's wave is the only writer
        const State s = unpack(leaf_packed);
        const Legal L = legal_actions(s, /*fallback_forced=*/0);     // python semantics (move_generator.py:24-70)
        const int n = legal_count(L);
        if (ROLE == 2) {                                       // the partner wave expands; this one only needs the value
            backup_value = n == 0 ? -1.0 : (double)value_ld;
            if (n == 0) split->nolegal_edge = leaf_edge; else split->wait_edge = leaf_edge;
        } else if (n == 0) {
            // portable_mcts.py:433-441: no legal move on a non-finished state => terminal, value -1
            backup_value = -1.0;
            if (lane == 0) {
                if (IS_ROOT) {
                    // (a live fresh root that turns out to have no move takes no chunk: its reservation ends here)
                    if (root_pending(t, g)) atomicSub(t.pool_stats + 2, 1);
                    nodes[0].nedges = 0; t.root_terminal[g] = 1; t.root_init_value[g] = -1.f;
                }
                else atomicOr(&edges[(size_t)leaf_edge].n_info, (uint32_t)kInfoTerminal << 24);   // value bits stay 0 (= -1)
            }
        } else {
            // gather per-lane logits / priors of the legal actions in ascending index order
            float h1 = 0.f, h2 = 0.f, hm = 0.f;
            const bool heads = priors220 == nullptr;
            if (heads && lane < kCells) {
                h1 = lp1[(size_t)g * 36 + lane]; h2 = lp2[(size_t)g * 36 + lane]; hm = lpm[(size_t)g * 36 + lane];
                if (tracing) {
                    float* th = t.trace_heads + tslot * 108;
                    th[lane] = h1; th[36 + lane] = h2; th[72 + lane] = hm;
                }
            }
            // Phase A, per 64 action indices (skipped when none of them is legal): legality, compact slot, logit.
            // The legal actions are then compacted through LDS so that Phase B (child state, terminal test, edge
            // record -- the expensive per-action work) runs once over min(n, 64) lanes instead of four times.
            float val[4]; int slot[4]; bool lg[4];
            int base = 0;
            float mx = -INFINITY;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int a = it * kWave + lane;
                lg[it] = a < 217 && legal_bit(L, a);
                val[it] = 0.f; slot[it] = 0;
                const uint64_t bal = __ballot(lg[it]);
                if (bal == 0) continue;                         // wave-uniform
                slot[it] = base + __popcll(bal & ((1ull << lane) - 1ull));
                base += __popcll(bal);
                int from = 0, dest = 0, cell = 0;
                if (a >= 36 && a < 180) { from = (a - 36) >> 2; dest = move_dest(from, (a - 36) & 3); dest = dest < 0 ? 0 : (dest > 35 ? 35 : dest); }
                else if (a >= 180 && a < 216) cell = a - 180;
                else if (a < 36) cell = a;
                float x;
                if (heads) {
                    const float p1d = __shfl(h1, dest), p2f = __shfl(h2, from), p1c = __shfl(h1, cell), pmc = __shfl(hm, cell);
                    x = a < 36 ? p1c : a < 180 ? (p2f + p1d) : a < 216 ? pmc : 0.f;
                } else {
                    x = lg[it] ? priors220[(size_t)g * 220 + a] : 0.f;
                }
                val[it] = x;
                if (heads && lg[it]) mx = fmaxf(mx, x);
            }
            if (heads) {
                // softmax over the legal set (portable_mcts.py:381-386), fp32
                mx = lzw::wave_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int it = 0; it < 4; ++it) { val[it] = lg[it] ? expf(val[it] - mx) : 0.f; sum += val[it]; }
                sum = lzw::wave_sum(sum);
#pragma unroll
                for (int it = 0; it < 4; ++it) val[it] = val[it] / sum;
            }
            if (tracing) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (lg[it]) t.trace_priors[tslot * 220 + it * kWave + lane] = val[it];
            }
            // root noise mix (portable_mcts.py:451-459)
            if (IS_ROOT && noise != nullptr && n > 1) {
                const float keep = (float)(1.0 - (double)epsilon);
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (lg[it]) val[it] = keep * val[it] + epsilon * noise[(size_t)g * noise_stride + slot[it]];
            }
            LZ_TSTAMP(g, 2)                                    // legal set, head gather, softmax
            // compaction: lane k (and k + 64) takes over the k-th legal action in ascending index order
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (lg[it]) { sc.val[slot[it]] = val[it]; sc.act[slot[it]] = it * kWave + lane; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float cval[2]; int cact[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = r * kWave + lane;
                cval[r] = k < n ? sc.val[k] : 0.f;
                cact[r] = k < n ? sc.act[k] : 0;
            }
            // renormalise with a sequential fp32 sum in ascending action order (== the oracle's order): scalar loop
            // of v_readlane + add, no further LDS round trip
            float psum = 0.f;
#pragma unroll 4
            for (int k = 0; k < n; ++k) psum += lzw::lane_bcast(k < kWave ? cval[0] : cval[1], k & 63);
            const bool bad = !(psum > 0.f) || !isfinite(psum);
            LZ_TSTAMP(g, 3)                                    // compaction + sequential renormalisation sum
            // node + edge allocation: the node from the game's bump counter, the run of n edges from the game's open
            // chunk of the pool, or from a new chunk when it does not fit there (runs never straddle chunks)
            int node_id = 0, e0 = 0;
            if (lane == 0) {
                e0 = ne_ld;
                const int off = e0 & (t.chunk - 1);
                if (off == 0 || off + n > t.chunk) e0 = take_chunk(t, g, IS_ROOT);
                if (e0 >= 0) {
                    t.n_edges[g] = e0 + n;
                    if (IS_ROOT) node_id = 0;
                    else {
                        node_id = nn_ld; t.n_nodes[g] = node_id + 1;
                        Edge& in = edges[(size_t)leaf_edge];
                        in.child = node_id; in.cbegin = e0; in.cn = (uint8_t)n;
                        nodes[node_id].state = leaf_packed;
                        nodes[node_id].parent = leaf_parent;
                    }
                    nodes[node_id].edge_begin = e0;
                    nodes[node_id].nedges = n;
                }
                if (IS_ROOT) t.root_init_value[g] = value_ld;
            }
            e0 = __builtin_amdgcn_readfirstlane(e0);
            node_id = __builtin_amdgcn_readfirstlane(node_id);
            // pool exhausted (counted in pool_stats[0]; the engine sizes the pool so that this does not happen): the leaf
            // stays unexpanded -- its value is still backed up, the next visit evaluates and tries again
            const int n_write = e0 >= 0 ? n : 0;
            LZ_TSTAMP(g, 4)                                    // allocation
            // Phase B: one pass (two only when a movement position has more than 64 legal moves)
            Edge* new_run = edges + (size_t)(e0 >= 0 ? e0 : 0);      // wave-uniform base + 32-bit lane offset
            for (int r = 0; r < (n_write > kWave ? 2 : 1); ++r) {
                const int k = r * kWave + lane;
                if (k >= n_write) continue;
                const int a = r == 0 ? cact[0] : cact[1];
                State c = s;
                int kd, p, q2, ex;
                index_to_code(s.phase, a, kd, p, q2, ex);
                apply_legal(c, kd, p, q2);
                uint8_t info = c.player < 0 ? kInfoWhite : 0;
                if (game_status(c) != 0) {
                    const int tv = (int)terminal_value_for_mover(c);
                    info |= kInfoTerminal | (uint8_t)((tv + 1) << 2);
                }
                Edge rec;
                rec.W = 0.0;
                rec.P = bad ? (1.0f / (float)n) : ((r == 0 ? cval[0] : cval[1]) / psum);
                rec.n_info = (uint32_t)info << 24;
                rec.child = -1;
                rec.cbegin = 0;
                rec.act = (uint8_t)a;
                rec.cn = 0;
                rec.pad[0] = rec.pad[1] = rec.pad[2] = rec.pad[3] = rec.pad[4] = rec.pad[5] = 0;
                new_run[k] = rec;
            }
            backup_value = (double)value_ld;
            LZ_TSTAMP(g, 5)                                    // child states, terminal tests, edge records
        }
    }
    if (IS_ROOT || ROLE == 1) return;
    // ---- backup along the path (portable_mcts.py:123-138), one lane per path entry ----
    // value added at offset j = v0 * (-1)^(#mover changes at offsets > j); the root gets the fully flipped value.
    if (plen > 0) {
        int flips_above = 0;                                    // mover changes at offsets above the current chunk
        for (int hi = plen; hi > 0; hi -= kWave) {
            const int lo = hi > kWave ? hi - kWave : 0;
            const int j = lo + lane;
            const bool in = j < hi;
            const uint32_t pe = !in ? 0u : (lo == 0 ? (uint32_t)path_entry : (uint32_t)path[j]);
            const uint64_t F = __ballot(in && (pe & kPathFlip) != 0u);
            const int above = in ? __popcll(F >> (lane + 1)) : 0;   // flips at offsets > j inside the chunk
            if (in) {
                Edge* e = &edges[(size_t)(pe & ~kPathFlip)];
                const double v = ((above + flips_above) & 1) ? -backup_value : backup_value;
                atomicAdd(&e->n_info, 1u);
                unsafeAtomicAdd(&e->W, v);
            }
            flips_above += __popcll(F);
        }
        root.visits += 1;
        if (lane == 0) {
            t.root_visits[g] = root.visits;
            t.root_W[g] = root_w + ((flips_above & 1) ? -backup_value : backup_value);
        }
        if (root_after != nullptr) root_after->visits = root.visits;
    }
    LZ_TSTAMP(g, 6)                                            // backup issued
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int st() { return hipGetLastError() == hipSuccess ? LZ_OK : LZ_ERR_LAUNCH; }

Tree make_tree(const LzTreeDesc* d) {
    Tree t;
    t.B = (int)d->num_games; t.node_cap = d->node_cap; t.chunk = d->edge_chunk; t.path_cap = d->path_cap;
    t.chunk_cap = d->chunk_cap;
    t.chunk_list = d->chunk_list; t.n_chunks = d->n_chunks; t.free_chunks = d->free_chunks; t.pool_top = d->pool_top;
    t.pool_stats = d->pool_stats;
    t.root_state = reinterpret_cast<const Packed*>(d->root_state);
    t.nodes = reinterpret_cast<Node*>(d->nodes);
    t.edges = reinterpret_cast<Edge*>(d->edges);
    t.n_nodes = d->n_nodes; t.n_edges = d->n_edges; t.root_visits = d->root_visits; t.root_W = d->root_w;
    t.root_init_value = d->root_init_value;
    t.path = d->path; t.path_len = d->path_len; t.leaf_kind = d->leaf_kind;
    t.leaf_state = reinterpret_cast<Packed*>(d->leaf_state); t.leaf_value = d->leaf_value;
    t.root_terminal = d->root_terminal; t.active = d->active;
    t.leaf_edge = d->leaf_edge; t.leaf_parent = d->leaf_parent;
    t.c_puct = d->exploration_weight;
    // OFF unless LZ_TREE_F32SEL=1: measured slower on the bench's random-init nets (near-flat priors: the two best
    // candidates of a level are usually within the margin, so the double path runs anyway, after the fp32 one)
    { const char* e = getenv("LZ_TREE_F32SEL"); t.fast_select = (e && e[0] == '1'); }
    const bool tr = d->trace_cap > 0 && d->trace_kind && d->trace_leaf && d->trace_heads && d->trace_priors && d->trace_value;
    t.trace_kind = tr ? d->trace_kind : nullptr;
    t.trace_leaf = tr ? reinterpret_cast<Packed*>(d->trace_leaf) : nullptr;
    t.trace_heads = tr ? d->trace_heads : nullptr;
    t.trace_priors = tr ? d->trace_priors : nullptr;
    t.trace_value = tr ? d->trace_value : nullptr;
    t.trace_cap = tr ? (int)d->trace_cap : 0;
    t.eval_count = d->eval_count;
    const bool live = d->live_state && d->live_row && d->live_count && d->live_count_cap > 0;
    t.live_state = live ? reinterpret_cast<Packed*>(d->live_state) : nullptr;
    t.live_row = live ? d->live_row : nullptr;
    t.live_count = live ? reinterpret_cast<unsigned long long*>(d->live_count) : nullptr;
    return t;
}
bool tree_ok(const LzTreeDesc* d) {
    return d && d->num_games >= 0 && d->node_cap >= 2 && d->path_cap >= 3 &&
           d->edge_chunk >= 128 && (d->edge_chunk & (d->edge_chunk - 1)) == 0 && d->chunk_cap >= 1 &&
           d->pool_chunks >= 1 && (d->pool_chunks + 1) * (int64_t)d->edge_chunk <= (int64_t)1 << 31 &&
           d->pool_chunks >= d->num_games &&                      // a chunk for every root (see take_chunk)
           d->chunk_list && d->n_chunks && d->free_chunks && d->pool_top && d->pool_stats &&
           d->root_state && d->nodes && d->edges && d->n_nodes && d->n_edges &&
           d->root_visits && d->root_w && d->root_init_value && d->path && d->path_len && d->leaf_kind &&
           d->leaf_state && d->leaf_value && d->root_terminal && d->leaf_edge && d->leaf_parent;
}

}  // namespace
