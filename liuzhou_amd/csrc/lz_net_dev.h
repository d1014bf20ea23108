// lz_net_dev.h -- device side of the fused policy + bucketed-value ResNet forward (see lz_net.hip for the design
// notes): tile maps, the MFMA GEMM core, head phases, and the two pieces a kernel is built from --
//   net_setup<C,S,W>(P, lds, ctx)  once per workgroup (cell geometry, zeroed LDS board, head parameters)
//   net_pass<C,S,W>(...)           the WHOLE network on S samples, from packed states / planes to log-probs + value
// Included by lz_net.hip (the stand-alone forward kernel) and lz_search.hip (the persistent search kernel, in which a
// workgroup alternates network passes with the tree step of the games it owns).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liuzhou_hip.h"
#include "lz_wave.h"

namespace {


typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// Weights and per-channel parameters are read through buffer resources: the address is
// {SGPR descriptor, SGPR byte offset (wave-uniform: layer / K step / tile), one VGPR lane offset}, so the
// unrolled K loops need no 64-bit per-lane pointer arithmetic and nothing loop-invariant to hoist and spill.
struct Rsrc { __amdgpu_buffer_rsrc_t w, f; };
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ h8 load_wfrag(__amdgpu_buffer_rsrc_t r, int lane_off, int byte_off) {
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, byte_off, 0));
}
__device__ __forceinline__ f4 load_f4(__amdgpu_buffer_rsrc_t r, int lane_float_off, int float_off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_float_off * 4, float_off * 4, 0));
}

constexpr int kHead = 64;      // policy / value head channels
constexpr int kMlp = 128;
constexpr int kBins = 101;
constexpr int kPool = 3 * kHead;

struct NetParams {
    const _Float16* wfrag;
    const float* fp;
    int layer_off[LZ_NET_MAX_LAYERS];   // offsets in halfs: stem, (conv1, conv2) x blocks, heads
    int hf_gw, hf_w1, hf_w2, hf_out;   // offsets in halfs of the head FC fragments (gpool_linear, fc1, fc2, 3 out convs)
    int wfrag_bytes, fparams_bytes;
    int debug_stop;             // diagnostic builds only: leave the pass after phase k (0 = run everything)
    const long long* n_dev;     // optional device-side batch size (<= the host-side capacity N)
    int blocks;
    // float-parameter offsets
    int stem_bias, blk0, trunk_a, trunk_b, head_bias, p_gwT, p_a2, p_b2, p_out, v_w1T, v_b1, v_w2T, v_b2;
};

// W = waves per workgroup.  8 waves x S samples fills a CU with ONE workgroup (LDS-limited); 4 waves x S/2 samples
// lets TWO independent workgroups share a CU, so one's store / barrier / head phases overlap the other's MFMAs.
template <int C, int S, int W = 8>
struct Cfg {
    static constexpr int NPOS = S * 36;
    static constexpr int NT = NPOS / 16;
    static constexpr int WAVES = W;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int PG = NT / 9;               // cell groups (4 or 2)
    static constexpr int CG = WAVES / PG;           // channel groups (2 or 4)
    static constexpr int CT = C / 16;               // 16-channel output tiles
    // 16-channel output tiles per wave.  2 with 8 waves (two waves per SIMD, 256 registers each); 4 for the 4-wave
    // workgroup of the 128-channel net (one wave per SIMD, 512 registers): every activation fragment read from LDS then
    // feeds 4 MFMAs instead of 2, which halves the LDS operand traffic of the trunk.
    static constexpr int CTW = CT / CG;
    static constexpr int WAVES_PER_SIMD = (WAVES == 4 && CTW == 4) ? 1 : 2;
    static constexpr int KB = C / 32;               // 32-channel K blocks
    static constexpr int KBLOG = (KB == 1) ? 0 : (KB == 2) ? 1 : 2;
    // bytes per cell row: an ODD number of 16-byte slots (9 or 17), part of the conflict-free operand layout below
    static constexpr int STRIDE = C * 2 + 16;
    // Zero-bordered board: cell (r,c) of sample s sits at row 58*s + 1 + 7*(r+1) + c; one shared zero column
    // between board rows and zero rows above/below make every 3x3 tap an in-bounds read at a CONSTANT byte
    // offset from the top-left neighbour -> the tap offset is folded into the ds_read immediate, no VALU.
    static constexpr int CELLS = 58;
    static constexpr int ACT_OFF = 0;
    static constexpr int ZERO_OFF = S * CELLS * STRIDE;
    // head scratch: fp16 rows for the MFMA B operand (16 sample columns), fp32 for scalar epilogues
    static constexpr int POOL_STRIDE = kPool * 2 + 16;              // 400 B per sample row
    static constexpr int HID_STRIDE = kMlp * 2 + 16;                // 272 B per sample row
    static constexpr int VL_STRIDE = 112;                           // floats per sample (101 bins padded)
    // Two aliased regions (lifetimes are separated by workgroup barriers): A = pooled rows | policy logits / value
    // logits (written only after the FC layer that consumed the pooled rows), B = g vector (policy) | fc1 hidden
    // rows (value).  Keeps a 4-wave workgroup under 80 KB so that two of them share a CU's 160 KB.
    static constexpr int POOL_OFF = ZERO_OFF + STRIDE;
    static constexpr int PLOG_OFF = POOL_OFF;
    static constexpr int PLOG_BYTES = (S * 432 > 16 * VL_STRIDE * 4) ? S * 432 : 16 * VL_STRIDE * 4;
    static constexpr int A_BYTES = (16 * POOL_STRIDE > PLOG_BYTES) ? 16 * POOL_STRIDE : PLOG_BYTES;
    static constexpr int G_OFF = POOL_OFF + A_BYTES;
    static constexpr int HID_OFF = G_OFF;
    static constexpr int B_BYTES = (16 * kHead * 4 > 16 * HID_STRIDE) ? 16 * kHead * 4 : 16 * HID_STRIDE;
    static constexpr int PAR_OFF = G_OFF + B_BYTES;
    static constexpr int LDS_BYTES = PAR_OFF + 5 * kHead * 4;
    static constexpr int HP = 8 / (CG * CTW);       // passes over the 8 head output tiles (2 or 1)
};

// ---- conflict-free LDS operand layout ------------------------------------------------------------------------
// A ds_read_b128 is served in four 16-lane groups, {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32
// (MI355X_MICROARCH.md, LDS): one LDS cycle per group if its 16 lanes touch 16 different 16-byte slots mod 256 B.
// The B operand of v_mfma_f32_16x16x32_f16 puts column (lane & 15) = a board cell and K chunk (lane >> 4) on a
// lane, so a group mixes 8 cells ("X": columns 0-3,12-15) at chunk q with 8 cells ("Y": columns 4-11) at chunk q+1.
//   * inside a row the four 16-byte chunks of a 32-channel K block sit 32 bytes apart (two K blocks interleaved):
//     slot(cell, q) = (STRIDE/16 * row(cell) + 2q + const) mod 16;
//   * a tile is not 16 consecutive cells but TWO board cells x 8 samples: the X columns hold one cell, the Y columns
//     another, for the same 8 samples.  Sample rows are 58 apart (58 * STRIDE/16 = 10 mod 16), so 8 samples of one
//     cell cover the 8 slots of one parity; the two cells of a pair sit on rows of opposite parity, so X slots and
//     Y slots (shifted by 2) tile all 16 slots, for either assignment of the two K chunks.
// All tiles are therefore perfect: 4 LDS cycles per read instead of 12 (64 channels) / 8 (128 channels) with
// consecutive cells.  Rows stay where the zero-bordered board puts them, so 3x3 taps remain constant offsets; only
// the cell <-> (tile, column) assignment is chosen, which no other phase sees (accumulators keep the same
// assignment through all layers).
// A wave owns 9 such tiles = one HALF board (18 cells) of 8 samples, described in a frame mirrored so that row 0 is
// always the outer edge: top-half and bottom-half waves then have the same out-of-board taps at the same tile
// index, and those tile-taps are not computed at all (tap_live).  The mirror costs one wave-uniform row step
// (+-7 rows) and three weight-row offsets in SGPRs; every other offset stays an instruction immediate.
__host__ __device__ constexpr int board_row(int n) {                // row index of board cell n = 36*sample + 6*r + c
    const int s = n / 36, p = n - s * 36;
    const int r = p / 6, c = p - r * 6;
    return s * 58 + 1 + 7 * (r + 1) + c;
}
__host__ __device__ constexpr int chunk_pos(int chunk) {            // 16-byte position of channel chunk (8 channels)
    return 2 * (chunk & 3) + ((chunk >> 2) & 1) + 8 * (chunk >> 3);
}
template <int C, int S>
struct TileMap { unsigned short cell[S * 36 / 16][16]; };
// Cell pairs of a half board in the MIRRORED frame (r' = distance from the outer board edge, c): {X cell, Y cell}.
// Both cells of a pair lie on rows of opposite parity ((r + c) odd vs even), which is what makes the two column
// sets land on complementary slots, and they share their out-of-board taps as far as 9 cells in 9 tiles allow:
// tiles 0-2 have no row towards the edge, tile 3 no left column, tile 4 no right column (tap_live below).
constexpr int kPairCell[9][2][2] = {{{0, 0}, {0, 5}}, {{0, 1}, {0, 2}}, {{0, 3}, {0, 4}}, {{1, 0}, {2, 0}}, {{1, 5}, {2, 5}},
                                    {{1, 1}, {1, 2}}, {{1, 3}, {1, 4}}, {{2, 1}, {2, 2}}, {{2, 3}, {2, 4}}};
template <int C, int S>
constexpr TileMap<C, S> make_tile_map() {
    constexpr int xcols[8] = {0, 1, 2, 3, 12, 13, 14, 15}, ycols[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    TileMap<C, S> t{};
    for (int pg = 0; pg < S / 4; ++pg) {                                // cell group = (8 samples) x (board half)
        const int half = pg & 1, s0 = (pg >> 1) * 8;
        for (int i = 0; i < 9; ++i)
            for (int k = 0; k < 8; ++k)
                for (int y = 0; y < 2; ++y) {
                    const int rm = kPairCell[i][y][0], c = kPairCell[i][y][1];
                    const int r = half ? 5 - rm : rm;
                    t.cell[pg * 9 + i][y ? ycols[k] : xcols[k]] = (unsigned short)((s0 + k) * 36 + r * 6 + c);
                }
    }
    return t;
}
// Is tap (mirrored row t: 0 = towards the outer edge, 1 = own row, 2 = towards the centre; dx: 0 = left .. 2 = right)
// inside the board for BOTH cells of tile i?  Dead tile-taps are skipped entirely: 15 of 81 per 3x3 conv.
__host__ __device__ constexpr bool tap_live(int i, int t, int dx) {
    return i < 3 ? t != 0 : i == 3 ? dx != 0 : i == 4 ? dx != 2 : true;
}
template <int C, int S>
constexpr bool tile_map_ok() {                                      // compile-time proof of the claim above
    constexpr int NT = S * 36 / 16;
    constexpr int m = ((C * 2 + 16) / 16) & 15;
    const TileMap<C, S> t = make_tile_map<C, S>();
    bool seen[S * 36] = {};
    for (int tile = 0; tile < NT; ++tile)
        for (int g = 0; g < 2; ++g) {                               // groups {X at q, Y at q+1} and {Y at q, X at q+1}
            bool slot[16] = {};
            for (int col = 0; col < 16; ++col) {
                const bool is_x = col < 4 || col >= 12;
                const int q = (is_x == (g == 0)) ? 0 : 1;
                const int sl = (m * board_row(t.cell[tile][col]) + 2 * q) & 15;
                if (slot[sl]) return false;
                slot[sl] = true;
                if (g == 0) { if (seen[t.cell[tile][col]]) return false; seen[t.cell[tile][col]] = true; }
            }
        }
    for (int n = 0; n < S * 36; ++n) if (!seen[n]) return false;
    for (int tile = 0; tile < NT; ++tile)                           // every skipped tap is out of board for all 16 cells
        for (int col = 0; col < 16; ++col) {
            const int p = t.cell[tile][col] % 36, r = p / 6, c = p % 6;
            const int sy = ((tile / 9) & 1) ? -1 : 1;
            for (int tr = 0; tr < 3; ++tr)
                for (int dx = 0; dx < 3; ++dx) {
                    const int rr = r + sy * (tr - 1), cc = c + dx - 1;
                    const bool inside = rr >= 0 && rr < 6 && cc >= 0 && cc < 6;
                    if (!tap_live(tile % 9, tr, dx) && inside) return false;
                }
        }
    return true;
}
static_assert(tile_map_ok<64, 16>() && tile_map_ok<128, 8>() && tile_map_ok<64, 8>(),
              "operand tiles must be bank-conflict free");
__constant__ const TileMap<64, 8> kTileMap64h = make_tile_map<64, 8>();
__constant__ const TileMap<64, 16> kTileMap64 = make_tile_map<64, 16>();
__constant__ const TileMap<128, 8> kTileMap128 = make_tile_map<128, 8>();
// The same map computed instead of read (net_setup: a __constant__ read is a global round trip at the head of every launch):
// i = tile % 9 is a compile-time constant where it is used, so the pair's two cells are immediates and `col` selects one.
__host__ __device__ constexpr int tile_cell_calc(int pg, int i, int col) {
    const bool is_x = col < 4 || col >= 12;
    const int k = is_x ? (col < 4 ? col : col - 8) : col - 4;
    const int rm = kPairCell[i][is_x ? 0 : 1][0], c = kPairCell[i][is_x ? 0 : 1][1];
    const int r = (pg & 1) ? 5 - rm : rm;
    return ((pg >> 1) * 8 + k) * 36 + r * 6 + c;
}
template <int C, int S>
constexpr bool tile_calc_ok() {
    const TileMap<C, S> t = make_tile_map<C, S>();
    for (int tile = 0; tile < S * 36 / 16; ++tile)
        for (int col = 0; col < 16; ++col)
            if (t.cell[tile][col] != tile_cell_calc(tile / 9, tile % 9, col)) return false;
    return true;
}
static_assert(tile_calc_ok<64, 16>() && tile_calc_ok<128, 8>() && tile_calc_ok<64, 8>(), "computed tile map == table");
template <int C, int S> __device__ __forceinline__ int tile_cell(int tile, int col);
template <> __device__ __forceinline__ int tile_cell<64, 16>(int tile, int col) { return kTileMap64.cell[tile][col]; }
template <> __device__ __forceinline__ int tile_cell<128, 8>(int tile, int col) { return kTileMap128.cell[tile][col]; }
template <> __device__ __forceinline__ int tile_cell<64, 8>(int tile, int col) { return kTileMap64h.cell[tile][col]; }

// byte address of 16-byte channel chunk `chunk` of board cell n (n = 36*sample + 6*r + c)
template <int C, int S>
__device__ __forceinline__ int act_addr(int n, int chunk) {
    using K = Cfg<C, S>;
    return K::ACT_OFF + board_row(n) * K::STRIDE + (chunk_pos(chunk) << 4);
}

// ---- the GEMM core: acc[9 cell tiles][4 channel tiles] += W(layer) * act --------------------------------
// Register budget (one wave per SIMD, 512 registers): 288 accumulators (residual stream + conv1 output),
// weight fragments double-buffered across K steps (2 x 4 x 4), activation fragments in a 2-deep ring
// that runs one cell tile ahead of the MFMAs.
template <int NW> using AccT = f4[9][NW];      // [cell tile][channel tile of the wave]

// Compile-time schedule of the K steps of one conv.  3x3: rows in the order {own, towards the centre, towards the
// edge} (the walk starts and ends on the cell's own row, where `base` points for the 1x1 convs and the stores), three
// columns per row, K::KB channel blocks per tap.  `row` is in the wave's mirrored frame.
template <int C, int S, bool TAPS9, bool STEM>
struct Steps {
    using K = Cfg<C, S>;
    static constexpr int KBS = STEM ? 1 : K::KB;                    // K blocks per tap
    static constexpr int N = TAPS9 ? 9 * KBS : K::KB;
    static constexpr int seq(int step) { return TAPS9 ? step / (3 * KBS) : 0; }
    static constexpr int row(int step) { return !TAPS9 ? 1 : seq(step) == 0 ? 1 : seq(step) == 1 ? 2 : 0; }
    static constexpr int dx(int step) { return TAPS9 ? (step / KBS) % 3 : 1; }
    static constexpr int kb(int step) { return TAPS9 ? step % KBS : step; }
    // LDS byte offset relative to the LEFT neighbour on the current row (`base`)
    static constexpr int lds_imm(int step) { return dx(step) * K::STRIDE + (chunk_pos(kb(step) * 4) << 4); }
    // weight byte offset relative to the start of the tap row (3x3) / of the layer (1x1), for CTN output tiles
    static constexpr int w_imm(int step, int ctn) { return (TAPS9 ? dx(step) * KBS + kb(step) : step) * ctn * 1024; }
    static constexpr int w_row_bytes(int ctn) { return TAPS9 ? 3 * KBS * ctn * 1024 : 0; }
    static constexpr bool live(int i, int step) { return !TAPS9 || tap_live(i, row(step), dx(step)); }
    // rows to move `base` by after `step` (in units of the wave's row step): own -> centre-side -> edge-side -> own
    static constexpr int shift(int step) {
        return !TAPS9 ? 0 : step + 1 == N ? 1 : seq(step + 1) == seq(step) ? 0 : seq(step) == 0 ? 1 : -2;
    }
};

// One K step, software-pipelined IN PLACE: tile i's activation fragment register is reloaded for the next K step
// right after its two MFMAs have been issued, so the LDS reads are spread over the whole step (one ds_read per
// two MFMAs) and overlap the matrix pipe instead of forming a separate read phase.  The counted lgkmcnt waits
// the compiler derives from this order leave 8 reads in flight.  Tile-taps that are out of board for the whole
// tile issue neither the MFMAs nor the read.
template <int C, int S, bool TAPS9, bool STEM, int STEP, int NW>
__device__ __forceinline__ void gemm_step(AccT<NW>& acc, const h8 (&A)[NW], h8 (&B)[9], const unsigned char* lds,
                                          int (&base)[9], int row_step) {
    using T = Steps<C, S, TAPS9, STEM>;
    constexpr bool more = STEP + 1 < T::N;
    constexpr int shift = T::shift(STEP);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (T::live(i, STEP)) {
#pragma unroll
            for (int j = 0; j < NW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j], B[i], acc[i][j], 0, 0, 0);
        }
        if (shift != 0) base[i] += shift * row_step;
#ifndef LZ_EXP_NO_BRELOAD   /* timing experiment only: without the LDS operand reloads the results are wrong */
        if (more && T::live(i, more ? STEP + 1 : STEP))
            B[i] = *reinterpret_cast<const h8*>(lds + base[i] + T::lds_imm(more ? STEP + 1 : STEP));
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int C, int S, bool TAPS9, bool STEM, int CTN, int STEP, int NSTEPS, int NW>
struct GemmSteps {
    static __device__ __forceinline__ void run(AccT<NW>& acc, h8 (&A0)[NW], h8 (&A1)[NW], h8 (&B)[9], __amdgpu_buffer_rsrc_t rw,
                                               const int (&wrow)[3], int lane16, const unsigned char* lds,
                                               int (&base)[9], int row_step) {
        using T = Steps<C, S, TAPS9, STEM>;
        // weight fragments (L2) are prefetched one K step ahead into the other register pair
        if (STEP + 1 < NSTEPS) {
            constexpr int nx = STEP + 1 < NSTEPS ? STEP + 1 : STEP;
#ifndef LZ_EXP_NO_ALOAD     /* timing experiment only: without the weight-fragment loads the results are wrong */
#pragma unroll
            for (int j = 0; j < NW; ++j) A1[j] = load_wfrag(rw, lane16, wrow[T::row(nx)] + T::w_imm(nx, CTN) + j * 1024);
#else
#pragma unroll
            for (int j = 0; j < NW; ++j) A1[j] = A0[j];
#endif
        }
        gemm_step<C, S, TAPS9, STEM, STEP, NW>(acc, A0, B, lds, base, row_step);
        GemmSteps<C, S, TAPS9, STEM, CTN, STEP + 1, NSTEPS, NW>::run(acc, A1, A0, B, rw, wrow, lane16, lds, base, row_step);
    }
};
template <int C, int S, bool TAPS9, bool STEM, int CTN, int NSTEPS, int NW>
struct GemmSteps<C, S, TAPS9, STEM, CTN, NSTEPS, NSTEPS, NW> {
    static __device__ __forceinline__ void run(AccT<NW>&, h8 (&)[NW], h8 (&)[NW], h8 (&)[9], __amdgpu_buffer_rsrc_t,
                                               const int (&)[3], int, const unsigned char*, int (&)[9], int) {}
};

// The same with the weight fragments fetched TWO K steps ahead (three register sets in rotation).  A K step of two waves
// per SIMD lasts ~290 ns, an L2 hit 300 - 500 ns: one step ahead leaves part of every load's latency on the critical
// path (scripts/micro/wfrag_ring.hip, profiles/r04_micro_wfrag_ring.json: the K loop of this kernel with loads one step
// ahead runs 67 % over its no-weight-traffic bound, two steps ahead AT the bound).
template <int C, int S, bool TAPS9, bool STEM, int CTN, int STEP, int NSTEPS, int NW>
struct GemmSteps2 {
    static __device__ __forceinline__ void run(AccT<NW>& acc, h8 (&A0)[NW], h8 (&A1)[NW], h8 (&A2)[NW], h8 (&B)[9],
                                               __amdgpu_buffer_rsrc_t rw, const int (&wrow)[3], int lane16,
                                               const unsigned char* lds, int (&base)[9], int row_step) {
        using T = Steps<C, S, TAPS9, STEM>;
        if (STEP + 2 < NSTEPS) {
            constexpr int nx = STEP + 2 < NSTEPS ? STEP + 2 : STEP;
#pragma unroll
            for (int j = 0; j < NW; ++j) A2[j] = load_wfrag(rw, lane16, wrow[T::row(nx)] + T::w_imm(nx, CTN) + j * 1024);
        }
        gemm_step<C, S, TAPS9, STEM, STEP, NW>(acc, A0, B, lds, base, row_step);
        GemmSteps2<C, S, TAPS9, STEM, CTN, STEP + 1, NSTEPS, NW>::run(acc, A1, A2, A0, B, rw, wrow, lane16, lds, base, row_step);
    }
};
template <int C, int S, bool TAPS9, bool STEM, int CTN, int NSTEPS, int NW>
struct GemmSteps2<C, S, TAPS9, STEM, CTN, NSTEPS, NSTEPS, NW> {
    static __device__ __forceinline__ void run(AccT<NW>&, h8 (&)[NW], h8 (&)[NW], h8 (&)[NW], h8 (&)[9], __amdgpu_buffer_rsrc_t,
                                               const int (&)[3], int, const unsigned char*, int (&)[9], int) {}
};
#ifndef LZ_NET_APF
#define LZ_NET_APF 1            /* weight fragments fetched this many K steps ahead (1 or 2) */
#endif

// Fully unrolled over the K steps (9 taps x C/32 blocks): every LDS read offset is an immediate and every
// weight address is {descriptor, scalar offset, lane offset}, so a step is <= 9 ds_read + 2 buffer_load +
// <= 18 MFMA and (on the two row changes and at the end) 9 address adds.
// `layer_half_off` = offset of the layer in halfs, ct0 = first output tile of this wave (both wave-uniform);
// `mirror` = the wave owns a bottom half board (its mirrored rows run against the weight rows).
// first two weight fragments of a layer (own row, left tap -- the same for both mirror states): issued a whole
// phase early (before the barrier / LDS store phase that precedes the conv) so their L2 latency never sits on the
// critical path
template <int C, int S, bool TAPS9, bool STEM, int CTN, int NW>
__device__ __forceinline__ void load_first_frags(__amdgpu_buffer_rsrc_t rw, int layer_half_off, int ct0, int lane,
                                                 h8 (&A0)[NW]) {
    using T = Steps<C, S, TAPS9, STEM>;
    const int wbyte = __builtin_amdgcn_readfirstlane(layer_half_off * 2 + ct0 * 1024) + T::w_row_bytes(CTN);
#pragma unroll
    for (int j = 0; j < NW; ++j) A0[j] = load_wfrag(rw, lane * 16, wbyte + j * 1024);
}

template <int C, int S, bool TAPS9, bool STEM, int CTN, int NW>
__device__ __forceinline__ void conv_gemm(AccT<NW>& acc, __amdgpu_buffer_rsrc_t rw, int layer_half_off, int ct0,
                                          const unsigned char* lds, int (&base)[9], int lane, h8 (&A0)[NW],
                                          bool mirror, int row_step) {
    using T = Steps<C, S, TAPS9, STEM>;
    const int wbyte = __builtin_amdgcn_readfirstlane(layer_half_off * 2 + ct0 * 1024);
    const int wrow[3] = {wbyte + (mirror ? 2 : 0) * T::w_row_bytes(CTN), wbyte + T::w_row_bytes(CTN),
                         wbyte + (mirror ? 0 : 2) * T::w_row_bytes(CTN)};
    const int lane16 = lane * 16;
    h8 A1[NW], B[9];
#if LZ_NET_APF == 2
    h8 A2[NW];
    if (T::N > 1) {
#pragma unroll
        for (int j = 0; j < NW; ++j) A1[j] = load_wfrag(rw, lane16, wrow[T::row(1)] + T::w_imm(1, CTN) + j * 1024);
    }
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if (T::live(i, 0)) B[i] = *reinterpret_cast<const h8*>(lds + base[i] + T::lds_imm(0));
#if LZ_NET_APF == 2
    GemmSteps2<C, S, TAPS9, STEM, CTN, 0, T::N, NW>::run(acc, A0, A1, A2, B, rw, wrow, lane16, lds, base, row_step);
#else
    GemmSteps<C, S, TAPS9, STEM, CTN, 0, T::N, NW>::run(acc, A0, A1, B, rw, wrow, lane16, lds, base, row_step);
#endif
}

// workgroup barrier that only orders LDS traffic: prefetched global loads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// relu(x) as fp16: convert first (packed RTNE converts), then one packed fp16 max per pair -- rounding is monotone and
// 0 is exact, so max(cvt(x), 0) == cvt(max(x, 0)); 4 VALU per 4 values instead of 6
__device__ __forceinline__ h4 relu_h4(f4 v) {
    const h4 o = __builtin_convertvector(v, h4);
    const h4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    return __builtin_elementwise_max(o, z);
}
// a*x + b on four lanes as packed FMAs (the per-channel affine feeds an fp16 rounding, contraction is harmless here)
__device__ __forceinline__ f4 fma4(f4 x, f4 a, f4 b) { return __builtin_elementwise_fma(x, a, b); }

__device__ __forceinline__ h4 to_h4(float a, float b, float c, float d) {
    h4 r;
    r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
    return r;
}

// write relu(scale*acc + shift) (per channel) as fp16 rows; `chan_base` = first channel of co tile 0
// per-channel parameters of this lane's 2 x 4 channels (issued early, consumed by store_act)
template <int NW>
__device__ __forceinline__ void load_chan_params(__amdgpu_buffer_rsrc_t rf, int float_off, int chan_base, int lane,
                                                 f4 (&v)[NW]) {
    const int sub = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NW; ++j) v[j] = load_f4(rf, sub, float_off + chan_base + j * 16);
}
// byte offset of the lane's store for channel tile j of the wave, relative to tile 0: tiles 0 / 1 sit 4 chunk positions
// (64 B) apart; tiles 2 / 3 (4 tiles per wave: chan_base is then a multiple of 64) are the interleaved K block, one
// position (16 B) further (chunk_pos)
__host__ __device__ constexpr int store_off(int j) { return (j & 1) * 64 + (j >> 1) * 16; }

// The store address of the lane's cell in tile i is derived from base[i] (the read address of its left
// neighbour + the lane's K chunk): row(cell) = base[i] + STRIDE - 32*(lane>>4); the lane writes channels
// chan_base + 16j + 4q .. +4 (q = lane>>4, chan_base a multiple of 32), i.e. chunk chan_base/8 + 2j + (q>>1), half
// (q&1).  One per-lane delta + the immediate 64j instead of 18 independently computed (hoisted, spilled) addresses.
template <int C, int S>
__device__ __forceinline__ int store_delta(int chan_base, int lane) {
    using K = Cfg<C, S>;
    const int q = lane >> 4;
    return K::STRIDE - 32 * q + (chunk_pos(chan_base >> 3) << 4) + 32 * (q >> 1) + 8 * (q & 1);
}

template <int C, int S, bool HAS_SCALE, int NW>
__device__ __forceinline__ void store_act(const AccT<NW>& acc, unsigned char* lds, const int (&base)[9], int chan_base,
                                          const f4 (&sc)[NW], const f4 (&sh)[NW], int lane) {
    const int delta = store_delta<C, S>(chan_base, lane);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            f4 v = acc[i][j];
            if (HAS_SCALE) v = fma4(v, sc[j], sh[j]); else v = v + sh[j];
            *reinterpret_cast<h4*>(lds + base[i] + delta + store_off(j)) = relu_h4(v);
        }
    }
}

// global pooling of a [cell][64] fp16 map in LDS -> fp16 row pooled[s][192] = mean | max | sqrt(var + 1e-6)
// (src/neural_network.py:67-80).  Two lanes per (sample, 4-channel group), 18 cells each, one pass
// (sum, sum of squares, max in fp32), halves combined with a quad-permute DPP swap.
template <int C, int S>
__device__ __forceinline__ void gpool64(unsigned char* lds, int tid) {
    using K = Cfg<C, S>;
    if (tid < S * 32) {
        const int pair = tid >> 1, half = tid & 1;
        const int s = pair >> 4, cq = pair & 15;
        // The lane's 18 cells are three board rows of six: consecutive cells sit STRIDE bytes apart, rows 7 * STRIDE, so
        // every read is the lane's base + an immediate (no per-cell index arithmetic); four channels at a time as packed
        // fp32 adds / multiplies, the maximum on the fp16 values themselves (exact).  Same additions in the same order
        // as a scalar loop over the cells: bit-identical sums.
        const unsigned char* bp = lds + K::ACT_OFF + (s * K::CELLS + 1 + 7 * (half * 3 + 1)) * K::STRIDE +
                                  (chunk_pos(cq >> 1) << 4) + (cq & 1) * 8;
        f4 sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
        const _Float16 ninf = (_Float16)(-INFINITY);
        h4 mxh = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const h4 v = *reinterpret_cast<const h4*>(bp + (7 * dr + c) * K::STRIDE);
                const f4 f = __builtin_convertvector(v, f4);
                sum = sum + f;
                sq = sq + f * f;
                mxh = __builtin_elementwise_max(mxh, v);
            }
        }
        const f4 mxf = __builtin_convertvector(mxh, f4);
        float mean[4], sd[4], mx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                              // combine with the neighbouring lane (tid ^ 1)
            const float su = sum[k] + lzw::dpp_f32<0xB1, 0xf>(0.f, sum[k]);       // quad_perm [1,0,3,2]
            const float qu = sq[k] + lzw::dpp_f32<0xB1, 0xf>(0.f, sq[k]);
            mx[k] = fmaxf(mxf[k], lzw::dpp_f32<0xB1, 0xf>(-INFINITY, mxf[k]));
            mean[k] = su * (1.0f / 36.0f);
            const float var = fmaxf(qu * (1.0f / 36.0f) - mean[k] * mean[k], 0.f);
            sd[k] = sqrtf(var + 1e-6f);
        }
        if (half == 0) {
            unsigned char* row = lds + K::POOL_OFF + s * K::POOL_STRIDE;
            *reinterpret_cast<h4*>(row + (cq * 4) * 2) = to_h4(mean[0], mean[1], mean[2], mean[3]);
            *reinterpret_cast<h4*>(row + (kHead + cq * 4) * 2) = to_h4(mx[0], mx[1], mx[2], mx[3]);
            *reinterpret_cast<h4*>(row + (2 * kHead + cq * 4) * 2) = to_h4(sd[0], sd[1], sd[2], sd[3]);
        }
    }
}

// small dense layer on the matrix cores: D[16 outputs of tile ct][16 samples] = W(ct) * rows, K = 32*kbn.
// In two parts, so that the weight fragments (L2 round trips) can be issued a phase ahead of the rows they multiply.
template <int CTN, int KBN>
__device__ __forceinline__ void fc_load(__amdgpu_buffer_rsrc_t rw, int half_off, int ct, int lane, h8 (&a)[KBN]) {
    const int wbyte = __builtin_amdgcn_readfirstlane(half_off * 2 + ct * 1024);
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) a[kb] = load_wfrag(rw, lane * 16, wbyte + kb * CTN * 1024);
}
template <int KBN>
__device__ __forceinline__ f4 fc_mma(const h8 (&a)[KBN], const unsigned char* rows, int row_stride, int lane) {
    f4 d = (f4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* bp = rows + (lane & 15) * row_stride + (lane >> 4) * 16;
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb)
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kb], *reinterpret_cast<const h8*>(bp + kb * 64), d, 0, 0, 0);
    return d;
}
template <int CTN, int KBN>
__device__ __forceinline__ f4 fc_tile(__amdgpu_buffer_rsrc_t rw, int half_off, int ct,
                                      const unsigned char* rows, int row_stride, int lane) {
    h8 a[KBN];
    fc_load<CTN, KBN>(rw, half_off, ct, lane, a);
    return fc_mma<KBN>(a, rows, row_stride, lane);
}

// write one head map (64 channels wide) from a wave's 2 output tiles: channel = (tile_in_map*16) + ...
template <int C, int S, int NW>
__device__ __forceinline__ void store_head(const AccT<NW>& acc, unsigned char* lds, const int (&base)[9], int map_tile0,
                                           __amdgpu_buffer_rsrc_t rf, int bias_off, int lane) {
    const int sub = (lane >> 4) * 4;
    const int delta = store_delta<C, S>(map_tile0 * 16, lane);
    f4 b[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) b[j] = load_f4(rf, sub, bias_off + (map_tile0 + j) * 16);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const f4 v = acc[i][j] + b[j];
            *reinterpret_cast<h4*>(lds + base[i] + delta + store_off(j)) = relu_h4(v);
        }
    }
}

// ---- per-workgroup context: everything a pass needs that does not change between passes -------------------------------
template <int C, int S, int W>
struct NetCtx {
    int tid, lane, wave, pg, cg, ct0, chan0;
    __amdgpu_buffer_rsrc_t rw, rf;
    int base[9];                                         // left neighbour of the lane's cell, + the lane's K chunk
    bool mirror;                                         // bottom half board: mirrored rows run upwards
    int row_step;
#ifdef LZ_EXP_HEAD_STAMPS
    uint64_t t_entry, t_setup;                           // timing experiment only
#endif
};

template <int C, int S, int W>
__device__ __forceinline__ void net_setup(const NetParams& P, unsigned char* lds, NetCtx<C, S, W>& ctx) {
    using K = Cfg<C, S, W>;
    static_assert(K::CT == K::CG * K::CTW && (K::CTW == 2 || K::CTW == 4), "each wave owns 2 or 4 output-channel tiles");
    static_assert(K::NT % 9 == 0 && K::PG * K::CG == K::WAVES, "waves = cell groups x channel groups");
    static_assert(S * 32 <= K::THREADS, "global pooling uses 2 lanes per (sample, 4-channel group)");
    constexpr int NTHR = K::THREADS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave % K::PG, cg = wave / K::PG;
    const int tile0 = pg * 9;
    ctx.tid = tid; ctx.lane = lane; ctx.wave = wave; ctx.pg = pg; ctx.cg = cg;
    ctx.ct0 = cg * K::CTW;                       // first output-channel tile of this wave
    ctx.chan0 = ctx.ct0 * 16;
    ctx.rw = make_rsrc(P.wfrag, P.wfrag_bytes);
    ctx.rf = make_rsrc(P.fp, P.fparams_bytes);
    // per-lane cell geometry of the 9 tiles
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int n = tile_cell_calc(pg, i, lane & 15);
        ctx.base[i] = K::ACT_OFF + board_row(n) * K::STRIDE + (lane >> 4) * 32 - K::STRIDE;
    }
    ctx.mirror = (pg & 1) != 0;
    ctx.row_step = ctx.mirror ? -7 * K::STRIDE : 7 * K::STRIDE;
    // zero the whole activation buffer once: the board borders stay zero for every layer / pass
    for (int i = tid; i < K::ZERO_OFF / 16; i += NTHR) reinterpret_cast<uint4*>(lds + K::ACT_OFF)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < (K::PAR_OFF - K::POOL_OFF) / 4; i += NTHR) reinterpret_cast<uint32_t*>(lds + K::POOL_OFF)[i] = 0u;
    // (no global read in here: the tile map is computed, the policy head's parameters are fetched by the pass that uses
    //  them -- a launch of one pass per workgroup starts with LDS writes only)
}

// One pass: samples n0 .. n0 + nvalid - 1 (nvalid <= S) of `packed` (32-byte bitboard records) or `planes`.
// Starts with a workgroup barrier (the previous user of the LDS buffers is done, and -- in the persistent search
// kernel -- the leaf states the tree step wrote are visible); ends without one.  Outputs go to global memory.
template <int C, int S, int W>
__device__ __forceinline__ void net_pass(const NetParams& P, unsigned char* lds, NetCtx<C, S, W>& ctx, const float* planes,
                                         const uint64_t* packed, int64_t n0, int nvalid, float* lp1, float* lp2,
                                         float* lpm, float* vlogits, float* value) {
    using K = Cfg<C, S, W>;
    constexpr int NW = K::CTW;
    using Acc = AccT<NW>;
    constexpr int NTHR = K::THREADS;
    const int tid = ctx.tid, lane = ctx.lane, wave = ctx.wave, cg = ctx.cg, ct0 = ctx.ct0, chan0 = ctx.chan0;
    const float* fp = P.fp;
    const __amdgpu_buffer_rsrc_t rw = ctx.rw, rf = ctx.rf;
    int (&base)[9] = ctx.base;
    const bool mirror = ctx.mirror;
    const int row_step = ctx.row_step;
    float* gvec = reinterpret_cast<float*>(lds + K::G_OFF);
    float* plog = reinterpret_cast<float*>(lds + K::PLOG_OFF);
    float* par = reinterpret_cast<float*>(lds + K::PAR_OFF);
    // ---- stage the 11 input planes as fp16 rows [cell][32 ch] (ch >= 11 zero) ----
    // (opaque copy of the thread id: the staging / head index arithmetic below is recomputed per pass instead of
    //  being hoisted out of the pass loop and spilled around the trunk)
    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));
    // packed records: 32 threads per sample, thread k of a sample stages cells k and k + 32 -- ONE 32-byte record per
    // thread, read before the barrier (it touches no LDS): its round trip overlaps the wait for the slowest wave
    const int st_s = tid_s >> 5, st_p = tid_s & 31;
    uint64_t rec0 = 0, rec1 = 0, rec2 = 0, rec3 = 0;
    if (packed != nullptr && st_s < nvalid) {
        const uint64_t* rec = packed + (n0 + st_s) * 4;
        rec0 = rec[0]; rec1 = rec[1]; rec2 = rec[2]; rec3 = rec[3];
    }
#ifdef LZ_EXP_HEAD_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const uint64_t t_before_barrier = __builtin_readcyclecounter();      // own loads and LDS writes have landed
#endif
    __syncthreads();
#ifdef LZ_EXP_HEAD_STAMPS
    const uint64_t t_after_barrier = __builtin_readcyclecounter();
    uint64_t fst[4] = {0, 0, 0, 0};
#define LZ_FSTAMP(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); fst[k] = __builtin_readcyclecounter(); }
#else
#define LZ_FSTAMP(k)
#endif
    LZ_FSTAMP(0)
    if (packed != nullptr) {
        if (st_s < S) {
            // 32-byte bitboard record (lz_rules.h: pack): planes = own, opp, own marks, opp marks, phase one-hot
            const bool on = st_s < nvalid;
            const bool white = (rec0 >> 53) & 1;
            const int phase = on ? (int)((rec0 >> 50) & 7) : 0;
            const uint64_t own = white ? rec1 : rec0, opp = white ? rec0 : rec1;
            const uint64_t sm = white ? rec3 : rec2, om = white ? rec2 : rec3;
            for (int p = st_p; p < 36; p += 32) {
                _Float16 row[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) row[k] = (_Float16)0.f;
                if (on) {
                    row[0] = (_Float16)(float)((own >> p) & 1);
                    row[1] = (_Float16)(float)((opp >> p) & 1);
                    row[2] = (_Float16)(float)((sm >> p) & 1);
                    row[3] = (_Float16)(float)((om >> p) & 1);
#pragma unroll
                    for (int ph = 1; ph <= 7; ++ph) row[3 + ph] = (_Float16)(phase == ph ? 1.f : 0.f);
                }
                const int n = st_s * 36 + p;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h8 v;
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = row[q * 8 + k];
                    *reinterpret_cast<h8*>(lds + act_addr<C, S>(n, q)) = v;
                }
            }
        }
    } else {
        for (int n = tid_s; n < K::NPOS; n += NTHR) {
            const int s = n / 36, p = n - s * 36;
            _Float16 row[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) row[k] = (_Float16)0.f;
            if (s < nvalid) {
                const float* src = planes + (n0 + s) * 396 + p;
#pragma unroll
                for (int ch = 0; ch < 11; ++ch) row[ch] = (_Float16)src[ch * 36];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                h8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = row[q * 8 + k];
                *reinterpret_cast<h8*>(lds + act_addr<C, S>(n, q)) = v;
            }
        }
    }
    LZ_FSTAMP(1)
    __syncthreads();
    LZ_FSTAMP(2)
    if (P.debug_stop == 1) return;                       // after input staging

    Acc x, acc;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) x[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 Af[NW];                                              // first weight fragments of the upcoming conv
    f4 pa[NW], pb[NW];                                       // per-channel parameters of the upcoming store
    // ---- stem: x = relu(conv(planes) + bias) ----
    load_first_frags<C, S, true, true, K::CT, NW>(rw, P.layer_off[0], ct0, lane, Af);
    load_chan_params<NW>(rf, P.stem_bias, chan0, lane, pb);
    conv_gemm<C, S, true, true, K::CT, NW>(x, rw, P.layer_off[0], ct0, lds, base, lane, Af, mirror, row_step);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            f4 v = x[i][j] + pb[j];
            x[i][j] = (f4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        }
    }
    LZ_FSTAMP(3)
    if (P.debug_stop == 2) { if (lane == 0 && x[0][0][0] == 123.f) lp1[0] = 1.f; return; }   // after the stem
    // ---- residual blocks: every global load (weights, parameters) is issued one phase ahead ----
#ifdef LZ_EXP_HEAD_STAMPS   /* timing experiment: stamps of every wave at the head sub-steps (scripts/exp_head_stamps.py) */
#define LZ_HSTAMP(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); hstamps[k] = __builtin_readcyclecounter(); }
    uint64_t hstamps[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#else
#define LZ_HSTAMP(k)
#endif
#ifdef LZ_EXP_STAMPS   /* timing experiment: s_memtime stamps of one wave through block 2 */
#define LZ_STAMP(k) if (blk == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamps[k] = __builtin_readcyclecounter(); }
    uint64_t stamps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#else
#define LZ_STAMP(k)
#endif
    for (int blk = 0; blk < P.blocks; ++blk) {
        const int bp = P.blk0 + blk * 3 * C;               // float offsets: a1 | b1 | bias1
        load_chan_params<NW>(rf, bp, chan0, lane, pa);
        load_chan_params<NW>(rf, bp + C, chan0, lane, pb);
        load_first_frags<C, S, true, false, K::CT, NW>(rw, P.layer_off[1 + 2 * blk], ct0, lane, Af);
        LZ_STAMP(0)
        lds_barrier();                                     // everyone finished reading the act buffer
        LZ_STAMP(1)
        store_act<C, S, true, NW>(x, lds, base, chan0, pa, pb, lane);                 // t = relu(a1*x + b1)
        load_chan_params<NW>(rf, bp + 2 * C, chan0, lane, pb);                          // bias1, used after conv1
        LZ_STAMP(2)
        lds_barrier();
        LZ_STAMP(3)
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        conv_gemm<C, S, true, false, K::CT, NW>(acc, rw, P.layer_off[1 + 2 * blk], ct0, lds, base, lane, Af, mirror, row_step);
        load_first_frags<C, S, true, false, K::CT, NW>(rw, P.layer_off[2 + 2 * blk], ct0, lane, Af);
        LZ_STAMP(4)
        lds_barrier();
        LZ_STAMP(5)
        store_act<C, S, false, NW>(acc, lds, base, chan0, pb, pb, lane);               // u = relu(conv1 + bias1)
        LZ_STAMP(6)
        lds_barrier();
        LZ_STAMP(7)
        conv_gemm<C, S, true, false, K::CT, NW>(x, rw, P.layer_off[2 + 2 * blk], ct0, lds, base, lane, Af, mirror, row_step);  // x += conv2(u)
        LZ_STAMP(8)
    }
#ifdef LZ_EXP_STAMPS
    if (blockIdx.x == 7 && (tid & 63) == 0 && vlogits != nullptr)
        for (int k = 0; k < 9; ++k) vlogits[wave * 16 + k] = (float)(stamps[k] - stamps[0]);
#endif
    if (P.debug_stop == 3) { if (lane == 0 && x[0][0][0] == 123.f) lp1[0] = 1.f; return; }   // after the trunk
    LZ_HSTAMP(0)
    // ---- trunk output h = relu(a*x + b) -> LDS; head 1x1 convs (8 output tiles: policy 0..3 | value 4..7) ----
    int tid_h = tid;
    asm volatile("" : "+v"(tid_h));                            // see tid_s: nothing below is live across the trunk
    const int lane_h = tid_h & 63;
    const int wh = P.layer_off[1 + 2 * P.blocks];
    const int ht0 = cg * K::CTW;                               // head tile of acc   (0..7)
    const int ht1 = K::CG * K::CTW + cg * K::CTW;              // head tile of x     (HP == 2 only)
    load_chan_params<NW>(rf, P.trunk_a, chan0, lane, pa);
    load_chan_params<NW>(rf, P.trunk_b, chan0, lane, pb);
    load_first_frags<C, S, false, false, 8, NW>(rw, wh, ht0, lane, Af);
    // bn2 scale / shift and the three output-conv rows of the policy head (5 x 64 floats): fetched here, parked in LDS
    // after the head convs, first read two barriers later
    constexpr int PPT = (5 * kHead + NTHR - 1) / NTHR;
    float parv[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid_h + j * NTHR;
        const int src = i < kHead ? P.p_a2 + i : i < 2 * kHead ? P.p_b2 + i - kHead : P.p_out + i - 2 * kHead;
        parv[j] = i < 5 * kHead ? fp[src] : 0.f;
    }
    lds_barrier();
    store_act<C, S, true, NW>(x, lds, base, chan0, pa, pb, lane);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) { acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; x[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; }
    // pass 0 -> acc, pass 1 (only when 4 waves' worth of tiles cover half of the 8 head tiles) -> x
    conv_gemm<C, S, false, false, 8, NW>(acc, rw, wh, ht0, lds, base, lane, Af, mirror, row_step);
    if (K::HP == 2) {
        load_first_frags<C, S, false, false, 8, NW>(rw, wh, ht1, lane, Af);
        conv_gemm<C, S, false, false, 8, NW>(x, rw, wh, ht1, lds, base, lane, Af, mirror, row_step);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j)
        if (tid_h + j * NTHR < 5 * kHead) par[tid_h + j * NTHR] = parv[j];
    __syncthreads();
    LZ_HSTAMP(1)
    if (P.debug_stop == 4) { if (lane == 0 && (acc[0][0][0] + x[0][0][0]) == 123.f) lp1[0] = 1.f; return; }   // after head convs
    // ---- policy head (skipped when the caller only wants values: lp1 == nullptr) ----
    if (lp1 != nullptr) {
    if (ht0 < 4) store_head<C, S, NW>(acc, lds, base, ht0, rf, P.head_bias, lane);
    // the weights of the next two steps travel while the map is pooled (every wave loads: an unconditional array is not
    // kept alive across the trunk, see the note on the head loads in DESIGN.md)
    h8 gw[6];
    fc_load<4, 6>(rw, P.hf_gw, wave & 3, lane_h, gw);
    const h8 wo0 = load_wfrag(rw, lane_h * 16, P.hf_out * 2);
    const h8 wo1 = load_wfrag(rw, lane_h * 16, P.hf_out * 2 + 1024);
    __syncthreads();
    LZ_HSTAMP(2)
    gpool64<C, S>(lds, tid_h);
    __syncthreads();
    LZ_HSTAMP(3)
    if (wave < 4) {                                            // g = gpool_linear(pooled): 4 tiles x K=192
        const f4 d = fc_mma<6>(gw, lds + K::POOL_OFF, K::POOL_STRIDE, lane_h);
        const int s = lane_h & 15, ch = wave * 16 + (lane_h >> 4) * 4;
        *reinterpret_cast<f4*>(gvec + s * kHead + ch) = d;
    }
    __syncthreads();
    LZ_HSTAMP(4)
    // three 1x1 output convs on p2 = relu(bn2(p + g)): p2 is formed in registers on the B fragment
    // (every [cell][8-channel chunk] is read exactly once), 1 output tile x K=64 on the matrix cores
    {
        // bn2's per-channel scale / shift of the lane's two 8-channel chunks, as vectors: the transform of a chunk is two
        // packed-fp32 adds, multiplies and adds and one packed fp16 max per four channels (the same operations per element
        // as the scalar form -- add, multiply, add, no contraction -- so the same values)
        f4 pa2[2][2], pb2[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int ch = (kb * 4 + (lane_h >> 4)) * 8 + hf * 4;
                pa2[kb][hf] = *reinterpret_cast<const f4*>(par + ch);
                pb2[kb][hf] = *reinterpret_cast<const f4*>(par + kHead + ch);
            }
        for (int t = wave; t < K::NT; t += K::WAVES) {
            f4 d = (f4){0.f, 0.f, 0.f, 0.f};
            const int n = t * 16 + (lane_h & 15);
            const int s = n / 36, p = n - s * 36;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int chunk = kb * 4 + (lane_h >> 4);
                const h8 v = *reinterpret_cast<const h8*>(lds + act_addr<C, S>(n, chunk));
                const f4 g0 = *reinterpret_cast<const f4*>(gvec + s * kHead + chunk * 8);
                const f4 g1 = *reinterpret_cast<const f4*>(gvec + s * kHead + chunk * 8 + 4);
                const h4 lo = __builtin_shufflevector(v, v, 0, 1, 2, 3), hi = __builtin_shufflevector(v, v, 4, 5, 6, 7);
                const h4 r0 = relu_h4((__builtin_convertvector(lo, f4) + g0) * pa2[kb][0] + pb2[kb][0]);
                const h4 r1 = relu_h4((__builtin_convertvector(hi, f4) + g1) * pa2[kb][1] + pb2[kb][1]);
                const h8 b = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb == 0 ? wo0 : wo1, b, d, 0, 0, 0);
            }
            if (lane_h < 16) {
                plog[(s * 3 + 0) * 36 + p] = d[0];
                plog[(s * 3 + 1) * 36 + p] = d[1];
                plog[(s * 3 + 2) * 36 + p] = d[2];
            }
        }
    }
    __syncthreads();
    LZ_HSTAMP(5)
    {   // log_softmax over the 36 cells of every (sample, head) row.  Sixteen lanes per row, nine of them with four
        // consecutive cells each: a wave normalises four rows at once with 4-step reductions inside the lane rows of 16
        // (cyclic rotations: every lane ends up with its row's maximum / sum), instead of one 36-of-64-lane row after the
        // other with 6-step whole-wave reductions -- half the instructions per wave.  (The sum of a row is now added up in
        // a different order than the whole-wave reduction did: last-bit differences in the log-probabilities.)
        constexpr int ROWS = S * 3, PASSES = (ROWS + 4 * K::WAVES - 1) / (4 * K::WAVES);
        const int k4 = lane_h & 15;
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int row = (pass * K::WAVES + wave) * 4 + (lane_h >> 4);
            const bool on = k4 < 9 && row < ROWS;
            const f4 v = on ? *reinterpret_cast<const f4*>(plog + row * 36 + 4 * k4) : (f4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            mx = fmaxf(mx, lzw::dpp_f32<0x128, 0xf>(mx, mx));       // row_ror:8
            mx = fmaxf(mx, lzw::dpp_f32<0x124, 0xf>(mx, mx));       // row_ror:4
            mx = fmaxf(mx, lzw::dpp_f32<0x122, 0xf>(mx, mx));       // row_ror:2
            mx = fmaxf(mx, lzw::dpp_f32<0x121, 0xf>(mx, mx));       // row_ror:1
            float e = 0.f;
            if (on) e = ((expf(v[0] - mx) + expf(v[1] - mx)) + expf(v[2] - mx)) + expf(v[3] - mx);
            e += lzw::dpp_f32<0x128, 0xf>(e, e);
            e += lzw::dpp_f32<0x124, 0xf>(e, e);
            e += lzw::dpp_f32<0x122, 0xf>(e, e);
            e += lzw::dpp_f32<0x121, 0xf>(e, e);
            const int s = row / 3, h = row - s * 3;
            if (on && s < nvalid) {
                const float lse = mx + logf(e);
                float* dst = (h == 0 ? lp1 : h == 1 ? lp2 : lpm) + (n0 + s) * 36 + 4 * k4;
                dst[0] = v[0] - lse; dst[1] = v[1] - lse; dst[2] = v[2] - lse; dst[3] = v[3] - lse;
            }
        }
    }
    __syncthreads();
    LZ_HSTAMP(6)
    }
    if (P.debug_stop == 5) { if (lane_h == 0 && x[0][0][0] == 123.f) lp1[0] = 1.f; return; }   // after the policy head
    // ---- value head ----
    if (K::HP == 2) store_head<C, S, NW>(x, lds, base, ht1 - 4, rf, P.head_bias + kHead, lane);
    else if (ht0 >= 4) store_head<C, S, NW>(acc, lds, base, ht0 - 4, rf, P.head_bias + kHead, lane);
    // fc1 / fc2 weights and biases of this wave's output tiles travel while the map is pooled (a tile index past the
    // last one is clamped: the load stays in bounds, the result is not used)
    constexpr int R1 = (8 + K::WAVES - 1) / K::WAVES, R2 = (7 + K::WAVES - 1) / K::WAVES;
    h8 w1[R1][6], w2[R2][4];
    f4 b1[R1], b2[R2];
#pragma unroll
    for (int r = 0; r < R1; ++r) {
        const int ct = wave + r * K::WAVES < 8 ? wave + r * K::WAVES : 7;
        fc_load<8, 6>(rw, P.hf_w1, ct, lane_h, w1[r]);
        b1[r] = load_f4(rf, (lane_h >> 4) * 4, P.v_b1 + ct * 16);
    }
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const int ct = wave + r * K::WAVES < 7 ? wave + r * K::WAVES : 6;
        fc_load<7, 4>(rw, P.hf_w2, ct, lane_h, w2[r]);
        b2[r] = load_f4(rf, (lane_h >> 4) * 4, P.v_b2 + ct * 16);       // bins past 101: never stored
    }
    __syncthreads();
    LZ_HSTAMP(7)
    gpool64<C, S>(lds, tid_h);
    __syncthreads();
    LZ_HSTAMP(8)
#pragma unroll
    for (int r = 0; r < R1; ++r) {                              // fc1 + relu: 8 tiles x K=192
        const int ct = wave + r * K::WAVES;
        if (ct < 8) {
            const f4 v = fc_mma<6>(w1[r], lds + K::POOL_OFF, K::POOL_STRIDE, lane_h) + b1[r];
            const int s = lane_h & 15, ch = ct * 16 + (lane_h >> 4) * 4;
            *reinterpret_cast<h4*>(lds + K::HID_OFF + s * K::HID_STRIDE + ch * 2) =
                to_h4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
        }
    }
    __syncthreads();
    LZ_HSTAMP(9)
    float* vl = plog;                                           // [16][112] value logits
#pragma unroll
    for (int r = 0; r < R2; ++r) {                              // fc2: 7 tiles (101 bins padded to 112) x K=128
        const int ct = wave + r * K::WAVES;
        if (ct < 7) {
            const f4 d = fc_mma<4>(w2[r], lds + K::HID_OFF, K::HID_STRIDE, lane_h);
            const int s = lane_h & 15, o = ct * 16 + (lane_h >> 4) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (o + q < kBins) vl[s * K::VL_STRIDE + o + q] = d[q] + b2[r][q];
        }
    }
    __syncthreads();
    LZ_HSTAMP(10)
    {   // bucket expectation, one wave per sample; a wave's samples side by side (see the log_softmax above)
        constexpr int SPW = (S + K::WAVES - 1) / K::WAVES;
        float v0[SPW], v1[SPW], mx[SPW], e0[SPW], e1[SPW], sum[SPW], ex[SPW];
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int s = wave + i * K::WAVES;
            const bool on = s < nvalid;
            v0[i] = on ? vl[s * K::VL_STRIDE + lane_h] : 0.f;
            v1[i] = (on && lane_h + 64 < kBins) ? vl[s * K::VL_STRIDE + lane_h + 64] : -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < SPW; ++i) mx[i] = lzw::wave_max(fmaxf(v0[i], v1[i]));
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            e0[i] = expf(v0[i] - mx[i]);
            e1[i] = lane_h + 64 < kBins ? expf(v1[i] - mx[i]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < SPW; ++i) sum[i] = lzw::wave_sum(e0[i] + e1[i]);
#pragma unroll
        for (int i = 0; i < SPW; ++i)
            ex[i] = lzw::wave_sum(e0[i] * (-1.0f + 0.02f * (float)lane_h) + e1[i] * (-1.0f + 0.02f * (float)(lane_h + 64)));
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int s = wave + i * K::WAVES;
            if (s < nvalid) {
                if (lane_h == 0 && value != nullptr) value[n0 + s] = ex[i] / sum[i];
                if (vlogits != nullptr) {
                    vlogits[(n0 + s) * kBins + lane_h] = v0[i];
                    if (lane_h + 64 < kBins) vlogits[(n0 + s) * kBins + lane_h + 64] = v1[i];
                }
            }
        }
    }
#ifdef LZ_EXP_HEAD_STAMPS
    LZ_HSTAMP(11)
    __syncthreads();
    if (blockIdx.x == 0 && lane_h == 0 && vlogits != nullptr) {        // over the block's own value logits (written above)
        for (int k = 0; k < 12; ++k) vlogits[wave * 24 + k] = (float)(hstamps[k] - hstamps[0]);
        vlogits[wave * 24 + 12] = (float)(ctx.t_setup - ctx.t_entry);   // net_setup
        vlogits[wave * 24 + 13] = (float)(fst[0] - ctx.t_setup);        // first barrier
        vlogits[wave * 24 + 14] = (float)(fst[1] - fst[0]);             // staging loop
        vlogits[wave * 24 + 15] = (float)(fst[2] - fst[1]);             // barrier after staging
        vlogits[wave * 24 + 16] = (float)(fst[3] - fst[2]);             // stem
        vlogits[wave * 24 + 17] = (float)(hstamps[0] - fst[3]);         // residual blocks
        vlogits[wave * 24 + 18] = (float)(t_before_barrier - ctx.t_setup);   // record load (hoisted before the barrier)
        vlogits[wave * 24 + 19] = (float)(t_after_barrier - t_before_barrier);   // the barrier itself
        vlogits[wave * 24 + 20] = (float)(ctx.t_entry & 0xFFFFFF);           // entry clock (low bits): start skew between waves
    }
#endif
}

// kernel-side view of a packed network (host): offsets copied from the descriptor; n_dev / debug_stop left off
inline NetParams make_net_params(const LzNetDesc* d) {
    NetParams P;
    P.wfrag = reinterpret_cast<const _Float16*>(d->wfrag);
    P.fp = d->fparams;
    for (int i = 0; i < LZ_NET_MAX_LAYERS; ++i) P.layer_off[i] = i < d->num_layers ? d->layer_offsets[i] : 0;
    P.blocks = d->blocks;
    P.n_dev = nullptr;
    P.wfrag_bytes = (int)d->wfrag_bytes; P.fparams_bytes = (int)d->fparams_bytes;
    P.debug_stop = 0;
    P.hf_gw = d->head_frag_offsets[0]; P.hf_w1 = d->head_frag_offsets[1]; P.hf_w2 = d->head_frag_offsets[2];
    P.hf_out = d->head_frag_offsets[3];
    P.stem_bias = d->off_stem_bias; P.blk0 = d->off_block0; P.trunk_a = d->off_trunk_a; P.trunk_b = d->off_trunk_b;
    P.head_bias = d->off_head_bias; P.p_gwT = d->off_p_gwT; P.p_a2 = d->off_p_a2; P.p_b2 = d->off_p_b2;
    P.p_out = d->off_p_out; P.v_w1T = d->off_v_w1T; P.v_b1 = d->off_v_b1; P.v_w2T = d->off_v_w2T; P.v_b2 = d->off_v_b2;
    return P;
}

}  // namespace
