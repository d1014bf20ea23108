"""One-off soak (GPU): the width-binned table-driven root bandit (octets / quadruples / pairs / singles, round 5) against the
plain IEEE-division kernel (LZ_ROOT_PUCT_DIV=1) on many random rows -- byte equality of visits, value sums and root values.
Rows: widths from the game's distribution and uniform ones, priors from softmaxes of varying sharpness (ties included),
leaf values in [-1, 1] with exact ties, zeros and +-1, exploration weights 0.25 .. 4, budgets 1 .. 20 000.
usage: python scripts/soak_bandit.py [rounds]   -> one JSON line"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liuzhou_amd import v0_core  # noqa: E402

DEV = "cuda:0"


def rows(rng, R, A):
    kind = rng.integers(0, 3)
    if kind == 0:
        w = np.clip(np.round(rng.gamma(2.2, 5.3, size=R)), 1, min(A, 60)).astype(np.int64)
    elif kind == 1:
        w = rng.integers(1, A + 1, size=R)
    else:
        w = rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, min(A, 64), A], size=R)
    valid = np.arange(A)[None, :] < np.minimum(w, A)[:, None]
    if rng.random() < 0.3:
        valid &= rng.random((R, A)) > 0.2
        valid[np.arange(R), 0] = True
    logits = rng.normal(size=(R, A)) * rng.choice([0.0, 0.5, 2.0, 6.0])
    if rng.random() < 0.3:
        logits = np.round(logits)                                  # many exact ties
    p = np.exp(logits - logits.max(1, keepdims=True)) * valid
    p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    leaf = rng.uniform(-1, 1, (R, A))
    if rng.random() < 0.4:
        leaf = np.round(leaf * 4) / 4
    return p, (leaf * valid).astype(np.float32), valid


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(20261004)
    total_rows = total_pulls = 0
    t0 = time.time()
    for i in range(rounds):
        A = int(rng.choice([8, 16, 24, 36, 64, 72, 130, 220]))
        R = int(rng.choice([1, 5, 333, 4096, 16384]))
        sims = int(rng.choice([1, 2, 17, 200, 1024, 4096, 20000])) if R <= 4096 else int(rng.choice([17, 200, 1024]))
        c = float(rng.choice([0.25, 1.0, 1.5, 4.0]))
        p, leaf, valid = rows(rng, R, A)
        args = [torch.from_numpy(x).to(DEV) for x in (p, leaf, valid)]
        os.environ.pop("LZ_ROOT_PUCT_DIV", None)
        fast = [t.cpu().numpy() for t in v0_core.root_puct_allocate_visits(*args, sims, c)]
        os.environ["LZ_ROOT_PUCT_DIV"] = "1"
        slow = [t.cpu().numpy() for t in v0_core.root_puct_allocate_visits(*args, sims, c)]
        os.environ.pop("LZ_ROOT_PUCT_DIV", None)
        for name, a, b in zip(("visits", "value_sum", "root_values"), fast, slow):
            if a.tobytes() != b.tobytes():
                bad = np.nonzero((a != b).reshape(R, -1).any(1))[0]
                print(json.dumps({"ok": False, "round": i, "what": name, "A": A, "R": R, "sims": sims, "c": c, "rows": bad[:8].tolist()}))
                sys.exit(1)
        assert (fast[0].sum(1) == sims).all()
        total_rows += R
        total_pulls += R * sims
    print(json.dumps({"ok": True, "rounds": rounds, "rows": total_rows, "pulls": total_pulls, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
