// lz_search.hip -- the whole search of one move as ONE kernel launch: a workgroup OWNS its games.
//
// Replaces the per-simulation launch pair of lz_tree_search (network kernel + tree kernel, i.e. two kernel
// boundaries and a device-wide barrier per simulation: 2 x (sims + 1) graph nodes per move) for the reference loop
//   v1/cpp/portable_mcts.cpp:483-590,832-939 / v1/python/portable_cpp_mcts.py:270-282
//   (prepare roots -> evaluate -> complete, then sims x (select -> evaluate -> complete)).
//
// Games never talk to each other, so nothing in that loop needs the whole grid:
//   * a workgroup of W waves owns S games for the whole move;
//   * per simulation it runs ONE network pass on its S pending leaf positions (lz_net_dev.h: the same code, tile
//     maps and MFMA order as the stand-alone kernel, so the evaluations are bit-identical to it) and then the tree
//     step of those S games (lz_tree_dev.h: expand + backup of simulation s, selection of simulation s + 1 -- the
//     same device functions the per-step kernels run), S / W games per wave;
//   * the only synchronisation is the workgroup barrier between the two phases; leaf states, head rows and values
//     are handed over through global memory (L2-resident, a few hundred bytes per game and simulation);
//   * with the 64-channel network two such workgroups (4 waves, 8 games, < 80 KB of LDS each) share a CU: while one
//     is in its latency-bound tree step the other one has the matrix pipes to itself.  Workgroups that land on a CU
//     second start half a period late (`stagger`), so the two do not run their phases in lockstep.
// Every wave runs exactly sims + 1 iterations and leaves: there is no work queue, no spinning, no grid barrier.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liuzhou_hip.h"
#include "lz_net_dev.h"
#include "lz_tree_dev.h"

namespace {

constexpr int kCuSlots = 4096;      // entries of the per-launch "which workgroup came second on this CU" table

// (XCC, SE, SH, CU) of the executing wave as a table index (HW_REG_HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13];
// HW_REG_XCC_ID: XCC_ID[3:0])
__device__ __forceinline__ int cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (8 << 6) | ((8 - 1) << 11));      // bits 8..15
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));   // bits 0..3
    return (int)(((xcc & 15u) << 8) | (hw & 255u)) & (kCuSlots - 1);
}

struct SearchArgs {
    int sims;
    const float* noise; int noise_stride; float epsilon;
    int* cu_slots;              // [kCuSlots] zeroed before the launch, or nullptr (no stagger)
    int stagger_ticks;          // delay of a CU's second workgroup, in 100 MHz ticks
    long long* phase_ticks;     // optional [grid][4]: 100 MHz ticks wave 0 spent in network passes / tree steps, CU slot, CU key
    int exp_mode;               // timing experiments (builds with -DLZ_EXP_SEARCH_MODES only; bit flags): 1 = no tree step and
                                // 2 = sleep instead of it (WRONG results), 4 = tree step at normal wave priority, 8 = priority
                                // edge alternating between the CU's two workgroups, 16 = younger workgroup favoured; 0 otherwise
};

template <int C, int S, int W>
__global__ __launch_bounds__(W * 64, 2) void tree_search_persistent_kernel(NetParams P, Tree t, SearchArgs a, float* lp1,
                                                                           float* lp2, float* lpm, float* values) {
    static_assert(S % W == 0, "every wave owns the same number of games");
    constexpr int GPW = S / W;                                  // games per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    NetCtx<C, S, W> ctx;
    net_setup<C, S, W>(P, lds, ctx);
    const int g0 = blockIdx.x * S;
    const int nvalid = (t.B - g0) < S ? (t.B - g0) : S;
    // ---- stagger: the second workgroup on a CU waits half a period so the pair alternates its phases ----
    int slot = 0;
    if (a.cu_slots != nullptr) {
        __shared__ int s_slot;
        if (threadIdx.x == 0) s_slot = atomicAdd(&a.cu_slots[cu_key()], 1);
        __syncthreads();
        slot = __builtin_amdgcn_readfirstlane(s_slot);
        if (a.stagger_ticks > 0 && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0 && (slot & 1)) {
            const uint64_t t0 = wall_clock64();
            while ((long long)(wall_clock64() - t0) < (long long)a.stagger_ticks) __builtin_amdgcn_s_sleep(64);
        }
        // the other waves wait for wave 0 at the barrier that opens the first pass
    }
    long long tk_net = 0, tk_tree = 0;
    for (int s = 0; s <= a.sims; ++s) {
        const uint64_t c0 = a.phase_ticks != nullptr ? wall_clock64() : 0;
        // ---- network pass on the S pending leaves (starts with a workgroup barrier) ----
        net_pass<C, S, W>(P, lds, ctx, nullptr, reinterpret_cast<const uint64_t*>(t.leaf_state), (int64_t)g0, nvalid, lp1,
                          lp2, lpm, nullptr, values);
        __syncthreads();                                       // head rows / values of all S games are visible
        // Unlike the per-step kernels (whose L1 starts empty at every launch), this kernel reads edge records in the selection
        // of simulation s that its own device atomics update in the backup of simulation s + 1.  The atomics execute at L2;
        // an agent-scope acquire drops whatever copy of those lines the CU's L1 may still hold from the previous simulation.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint64_t c1 = a.phase_ticks != nullptr ? wall_clock64() : 0;
        // ---- tree step of the games this wave owns ----
        if (!(a.exp_mode & 4)) __builtin_amdgcn_s_setprio(3);   // short bursts between dependent loads go first
        int wave_s = ctx.wave;
        asm volatile("" : "+s"(wave_s));                        // nothing of the tree step is hoisted across the pass
        // expand scratch: the fc1 hidden rows / g vectors of the pass are dead until the next pass stages its input
        static_assert(W * kExpandScratchBytes <= Cfg<C, S, W>::B_BYTES, "expand scratch must fit the dead head region");
        const ExpandScratch sc = {reinterpret_cast<float*>(lds + Cfg<C, S, W>::G_OFF + wave_s * kExpandScratchBytes),
                                  reinterpret_cast<int*>(lds + Cfg<C, S, W>::G_OFF + wave_s * kExpandScratchBytes + 320)};
        if (a.exp_mode & 2) {
            const uint64_t t0 = wall_clock64();
            while ((long long)(wall_clock64() - t0) < 2400) __builtin_amdgcn_s_sleep(16);
        }
#pragma unroll 1
        for (int j = 0; j < GPW; ++j) {
            const int g = g0 + wave_s * GPW + j;
            if (g >= t.B || (a.exp_mode & 3)) break;
            RootInfo root;
            if (s == 0) {
                tree_expand<true>(t, g, ctx.lane, lp1, lp2, lpm, nullptr, values, a.noise, a.noise_stride, a.epsilon, sc, &root, s);
                if (a.sims > 0) {
                    __threadfence_block();
                    root = load_root_info(t, g);                // the root record itself was just written
                    tree_select(t, g, ctx.lane, root);
                }
            } else {
                tree_expand<false>(t, g, ctx.lane, lp1, lp2, lpm, nullptr, values, nullptr, 0, 0.f, sc, &root, s);
                if (s < a.sims) {
                    __threadfence_block();
                    tree_select(t, g, ctx.lane, root);
                }
            }
        }
        // The instruction arbiter favours the OLDER wave: of the two workgroups on a CU the first to arrive would run its
        // passes ~20 % faster than the second and leave it alone on the CU at the end.  Alternating a one-step priority
        // edge between the pair, pass by pass, gives both the same average speed (they finish together).
        if ((a.exp_mode & 8) ? (((slot + s) & 1) != 0) : ((a.exp_mode & 16) ? ((slot & 1) != 0) : false))
            __builtin_amdgcn_s_setprio(1);
        else
            __builtin_amdgcn_s_setprio(0);
        if (a.phase_ticks != nullptr) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const uint64_t c2 = wall_clock64();
            tk_net += (long long)(c1 - c0); tk_tree += (long long)(c2 - c1);
        }
        // the next pass opens with a workgroup barrier: every wave's new leaf states are written by then
    }
    if (a.phase_ticks != nullptr && threadIdx.x == 0) {
        a.phase_ticks[4 * blockIdx.x] = tk_net;
        a.phase_ticks[4 * blockIdx.x + 1] = tk_tree;
        a.phase_ticks[4 * blockIdx.x + 2] = slot;
        a.phase_ticks[4 * blockIdx.x + 3] = cu_key();
    }
}

__global__ void zero_i32_kernel(int* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}

template <int C, int S, int W>
int configure_search() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(tree_search_persistent_kernel<C, S, W>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<C, S, W>::LDS_BYTES) == hipSuccess
               ? LZ_OK : LZ_ERR_LAUNCH;
}

bool g_search_configured[64] = {};          // per device: the dynamic-LDS attribute belongs to the device's code object

}  // namespace

// live timing hooks of lz_net.hip (bench.py's roofline probe): the search kernel is bracketed like a network launch
extern "C" int lz_prof_mark_begin(void* stream);
extern "C" int lz_prof_mark_end(void* stream, int64_t evals);

extern "C" {

int lz_tree_search_persistent(const LzTreeDesc* d, const LzNetDesc* net, int64_t sims, float* lp1, float* lp2,
                              float* lpmc, float* values, const float* noise, int64_t noise_stride, float epsilon,
                              int continue_trees, int32_t* cu_slots, int64_t stagger_us, int64_t* phase_ticks,
                              void* stream) {
    if (!tree_ok(d) || !net || sims < 0 || !lp1 || !lp2 || !lpmc || !values || stagger_us < 0) return LZ_ERR_ARG;
    if (!net->wfrag || !net->fparams) return LZ_ERR_ARG;
    if (net->blocks < 0 || net->blocks > 15 || net->num_layers != 2 + 2 * net->blocks) return LZ_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(net->wfrag) & 15) || (reinterpret_cast<uintptr_t>(net->fparams) & 15)) return LZ_ERR_ALIGN;
    // built for the 64-channel network (two 4-wave workgroups per CU); the fp32 parity mode and the 128-channel
    // network (one workgroup fills a CU's LDS: nothing would overlap the tree step) stay on lz_tree_search
    if (net->channels != 64 || (net->flags & (4 | 8))) return LZ_ERR_UNSUPPORTED;
    const int64_t B = d->num_games;
    if (B == 0) return LZ_OK;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return LZ_ERR_LAUNCH;
    if (!g_search_configured[device]) {
        const int rc = configure_search<64, 8, 4>();
        if (rc) return rc;
        g_search_configured[device] = true;
    }
    int rc = continue_trees ? LZ_OK : lz_tree_begin(d, stream);
    if (rc) return rc;
    hipStream_t stm = as_stream(stream);
    if (cu_slots != nullptr)
        hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(256), 0, stm, cu_slots, kCuSlots);
    const NetParams P = make_net_params(net);
    SearchArgs a;
    a.sims = (int)sims; a.noise = noise; a.noise_stride = (int)noise_stride; a.epsilon = epsilon;
    a.cu_slots = cu_slots;
    a.stagger_ticks = (int)((stagger_us > 10000 ? 10000 : stagger_us) * 100);     // 100 MHz ticks; clamped to 10 ms
    a.phase_ticks = reinterpret_cast<long long*>(phase_ticks);
    a.exp_mode = 0;
#ifdef LZ_EXP_SEARCH_MODES  /* experiment builds only (scripts/exp_persistent.py, profiles/r03_experiments.md): modes that skip
                             * or replace the tree step give WRONG results and must not be reachable in the shipped library */
    a.exp_mode = getenv("LZ_EXP_SEARCH_MODE") ? atoi(getenv("LZ_EXP_SEARCH_MODE")) : 0;
#endif
    using K = Cfg<64, 8, 4>;
    const unsigned grid = (unsigned)((B + 8 - 1) / 8);
    (void)lz_prof_mark_begin(stream);
    hipLaunchKernelGGL((tree_search_persistent_kernel<64, 8, 4>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, stm, P,
                       make_tree(d), a, lp1, lp2, lpmc, values);
    (void)lz_prof_mark_end(stream, B * (sims + 1));
    return st();
}

int lz_tree_search_persistent_grid(int64_t num_games) { return (int)((num_games + 7) / 8); }

}  // extern "C"
