// lz_soa.h -- the reference's 12-tensor struct-of-arrays state layout <-> bitboards (host + device).
// Layout contract: v0/include/v0/tensor_state_batch.hpp:11-35, v1/python/mcts_gpu.py:40-57.
#pragma once
#include "lz_rules.h"
#include "../../include/liuzhou_hip.h"

namespace lz {

// cells of a 36-byte row whose int8 value equals v
LZ_HD uint64_t cells_equal(const int8_t* row, int v) {
    uint64_t m = 0;
#pragma unroll
    for (int i = 0; i < kCells; ++i) m |= (uint64_t)((int)row[i] == v) << i;
    return m;
}
LZ_HD uint64_t cells_nonzero(const uint8_t* row) {
    uint64_t m = 0;
#pragma unroll
    for (int i = 0; i < kCells; ++i) m |= (uint64_t)(row[i] != 0) << i;
    return m;
}

// Raw view of one SoA row: the five value-sets the reference compares against, plus counters.
struct RawState {
    uint64_t black, white, empty, mb, mw;
    int64_t phase, player, pm_req, pm_rem, pc_req, pc_rem, forced, move_count, msc;
};

LZ_HD int clampi(int64_t v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

LZ_HD uint64_t pick_cells(const RawState& r, int64_t v) {
    return v == 1 ? r.black : v == -1 ? r.white : v == 0 ? r.empty : 0ull;
}

}  // namespace lz
