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
