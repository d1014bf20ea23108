"""Helpers to load the committed golden fixtures (tests/golden/*.npz, made by oracle/gen_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("board", "marks_black", "marks_white", "phase", "current_player",
          "pending_marks_required", "pending_marks_remaining",
          "pending_captures_required", "pending_captures_remaining",
          "forced_removals_done", "move_count", "moves_since_capture")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def states(z, prefix):
    return {f: z[f"{prefix}_{f}"] for f in FIELDS}


def unpack_mask(packed, width):
    return np.unpackbits(packed, axis=1)[:, :width].astype(bool)


def states_equal(a, b):
    for f in FIELDS:
        x = np.asarray(a[f]); y = np.asarray(b[f])
        if not np.array_equal(x.reshape(x.shape[0], -1).astype(np.int64), y.reshape(y.shape[0], -1).astype(np.int64)):
            return False, f
    return True, None


# ---- row hashes: how the reference-scale fixtures (g15 / g16 / g17) store wide per-row outputs ------------------------
def row_hash64(a):
    """FNV-1a (64 bit) over the little-endian int16 image of every row of `a` ([N, ...] integers that fit int16):
    uint64[N].  The generator (oracle/gen_golden_large.py) stores these instead of e.g. int32[N, 220, 4] metadata."""
    a = np.asarray(a)
    n = a.shape[0]
    flat = a.reshape(n, -1).astype(np.int64)
    if flat.size and (flat.min() < -32768 or flat.max() > 32767):
        raise ValueError("row_hash64: values do not fit int16")
    cols = np.ascontiguousarray(flat.astype("<i2")).view(np.uint8).reshape(n, -1)
    h = np.full(n, 0xCBF29CE484222325, np.uint64)
    prime = np.uint64(0x100000001B3)
    with np.errstate(over="ignore"):
        for j in range(cols.shape[1]):
            h = (h ^ cols[:, j].astype(np.uint64)) * prime
    return h


def state_rows(st):
    """The 12 state fields of a batch as one int16-able matrix [N, 36*3 + 9] (field order = FIELDS)."""
    n = np.asarray(st["board"]).shape[0]
    return np.concatenate([np.asarray(st[f]).reshape(n, -1).astype(np.int64) for f in FIELDS], axis=1)


def group_hash64(item_hash, group, num_groups):
    """Order-sensitive combination of the items' hashes per group (items of a group are consecutive, `group` ascending):
    sum_k item_hash[k] * odd_weight(rank of k inside its group) mod 2^64 -> uint64[num_groups] (0 for empty groups)."""
    item_hash = np.asarray(item_hash, np.uint64)
    group = np.asarray(group, np.int64)
    out = np.zeros(num_groups, np.uint64)
    if item_hash.size == 0:
        return out
    assert np.all(np.diff(group) >= 0), "items must be grouped"
    starts = np.flatnonzero(np.r_[True, np.diff(group) != 0])
    first = np.repeat(starts, np.diff(np.r_[starts, group.size]))
    rank = (np.arange(group.size) - first).astype(np.uint64)
    with np.errstate(over="ignore"):
        w = (rank * np.uint64(2) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        out[group[starts]] = np.add.reduceat(item_hash * w, starts)
    return out
