#!/usr/bin/env python3
"""Reference-scale regression vectors: the reference's own suites are 10 000 synthetic masks
(tests/v0/cuda/test_fast_legal_mask_cuda.py:74-118,179-228, seed 0xF00DCAFE), 10 000 applies
(tests/v0/cuda/test_fast_apply_moves_cuda.py:127-247,343-373, seed 0xA11CEB0B) and 4 000 + 1 000 playout states
(tests/v0/test_actions.py:115-159, seed 0x7777).  This script produces fixtures of that size FROM THE REFERENCE:

  g15_rules_large.npz    >= 5 000 reachable non-terminal states of src/ random playouts (seed 0x7777), their legal sets
                         (bits), the reference operator's metadata (per-row hash) and every child transition of the
                         Python rule engine (per-state hash of the children in ascending action order; the reference's
                         CPU operators are checked against the Python engine while generating)
  g16_garbage_large.npz  the 10 000 synthetic states of the reference mask test, drawn by ITS OWN generator function
                         under ITS seed, -> mask bits + per-row metadata hash from the reference CPU operator, T = 217 / 220
  g17_apply_micro.npz    10 000 per-kind micro-positions + action codes drawn by the reference apply test's own generator
                         under its seed -> the reference CPU operator's 12 output tensors

Only data is stored (inputs and expected outputs); masks as bits, metadata / children as 64-bit row hashes
(tests/golden_utils.py: row_hash64 / group_hash64), so the three files stay well under 2 MB.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_large.py
"""
from __future__ import annotations

import importlib.util
import os
import random
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np
import torch

import gen_golden as G                      # reference imports + helpers (pack_states, legal_index_list, ...)

# the row-hash helpers the tests use (loaded by path: the reference has a `tests` package of its own on sys.path)
_spec = importlib.util.spec_from_file_location("lz_golden_utils", os.path.join(REPO, "tests", "golden_utils.py"))
_gu = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gu)
FIELDS, group_hash64, row_hash64, state_rows = _gu.FIELDS, _gu.group_hash64, _gu.row_hash64, _gu.state_rows

v0_core = G.v0_core
OUT = G.OUT
REF = G.REF


def load_reference_test(rel):
    """Import one of the reference's test modules by path (its generator functions and seeds are what we want)."""
    path = os.path.join(REF, rel)
    spec = importlib.util.spec_from_file_location("ref_" + os.path.basename(rel)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_rules_large():
    t0 = time.time()
    nonterm, _term = G.random_playout_states(640, 0x7777)
    rng = random.Random(1)
    by_phase = {}
    for s in nonterm:
        by_phase.setdefault(s.phase, []).append(s)
    P = G.Phase
    quota = {P.PLACEMENT: 1300, P.MARK_SELECTION: 900, P.REMOVAL: 320, P.MOVEMENT: 2300, P.CAPTURE_SELECTION: 900,
             P.FORCED_REMOVAL: 320, P.COUNTER_REMOVAL: 320}
    chosen = []
    for p, lst in by_phase.items():
        rng.shuffle(lst)
        chosen.extend(lst[: quota.get(p, 100)])
    print("[g15] reachable by phase:", {p.name: len(v) for p, v in by_phase.items()}, "chosen", len(chosen),
          f"({time.time() - t0:.0f} s)")
    st = G.pack_states(chosen)
    masks = np.zeros((len(chosen), G.TOTAL_DIM), bool)
    child_states, child_parent, child_action = [], [], []
    for i, s in enumerate(chosen):
        moves, idx = G.legal_index_list(s)
        for k in np.argsort(idx):
            masks[i, idx[k]] = True
            child_states.append(G.apply_move(s, moves[k], quiet=True))
            child_parent.append(i)
            child_action.append(idx[k])
    ch = G.pack_states(child_states)
    child_parent = np.array(child_parent, np.int64)
    child_action = np.array(child_action, np.int64)
    # the reference's own tensor operators agree with its Python engine on all of this
    m_ref, meta_ref = v0_core.encode_actions_fast(*G.to_torch(st)[:10], 36, 144, 36, 4)
    assert np.array_equal(m_ref.numpy(), masks), "reference v0_core mask != reference python"
    meta = meta_ref.numpy()
    codes = meta[child_parent, child_action]
    applied = v0_core.batch_apply_moves(*G.to_torch(st), torch.from_numpy(codes.copy()), torch.from_numpy(child_parent))
    for f, t in zip(FIELDS, applied):
        assert np.array_equal(t.numpy().astype(ch[f].dtype), ch[f]), f"reference apply mismatch in {f}"
    np.savez_compressed(
        os.path.join(OUT, "g15_rules_large.npz"),
        legal_mask=np.packbits(masks, axis=1),
        metadata_hash=row_hash64(meta),
        children_hash=group_hash64(row_hash64(state_rows(ch)), child_parent, len(chosen)),
        num_children=np.int64(len(child_states)),
        **G.prefixed("s", st))
    print(f"[g15] states={len(chosen)} transitions={len(child_states)} ({time.time() - t0:.0f} s)")


def gen_garbage_large():
    mod = load_reference_test("tests/v0/cuda/test_fast_legal_mask_cuda.py")
    n = int(getattr(mod, "NUM_STATES", 10000))
    torch.manual_seed(int(mod.SEED))                       # what the test's first draw sees
    batch = [t.clone() for t in mod._random_cpu_batch(n)]
    st = {f: x.numpy() for f, x in zip(FIELDS[:10], batch)}
    st["move_count"] = np.zeros(n, np.int64)
    st["moves_since_capture"] = np.zeros(n, np.int64)
    out = {}
    for aux, key in ((1, "t217"), (4, "t220")):
        m, meta = v0_core.encode_actions_fast(*batch, 36, 144, 36, aux)
        out[f"mask_{key}"] = np.packbits(m.numpy(), axis=1)
        out[f"meta_hash_{key}"] = row_hash64(meta.numpy())
        print(f"[g16] T={216 + aux}: legal per state avg={m.sum(1).float().mean():.2f}")
    planes = v0_core.states_to_model_input(*batch[:5])
    out["model_input_hash"] = row_hash64(planes.numpy().astype(np.int8))
    np.savez_compressed(os.path.join(OUT, "g16_garbage_large.npz"), seed=np.int64(mod.SEED), **out, **G.prefixed("s", st))
    print(f"[g16] garbage states={n} (reference generator, seed 0x{int(mod.SEED):X})")


def gen_apply_micro():
    mod = load_reference_test("tests/v0/cuda/test_fast_apply_moves_cuda.py")
    n = int(getattr(mod, "NUM_ACTIONS", 10000))
    mod.random_rng.seed(int(mod.SEED))                      # what the test's first draw sees
    torch.manual_seed(int(mod.SEED))
    batch = mod._random_apply_batch(n)
    st = {f: x.numpy() for f, x in zip(FIELDS, batch[:12])}
    codes = batch[12].numpy().astype(np.int32)
    out = v0_core.batch_apply_moves(*batch)
    ch = {f: t.numpy() for f, t in zip(FIELDS, out)}
    kinds = np.bincount(codes[:, 0], minlength=9)
    changed = int((row_hash64(state_rows(ch)) != row_hash64(state_rows(st))).sum())
    np.savez_compressed(os.path.join(OUT, "g17_apply_micro.npz"), seed=np.int64(mod.SEED), codes=codes.astype(np.int8),
                        **G.prefixed("s", st), **G.prefixed("c", ch))
    print(f"[g17] micro-positions={n} (reference generator, seed 0x{int(mod.SEED):X}); per action kind {kinds.tolist()}; "
          f"{changed} rows changed by the move")


def main():
    which = set(sys.argv[1:])
    if not which or "g15" in which:
        gen_rules_large()
    if not which or "g16" in which:
        gen_garbage_large()
    if not which or "g17" in which:
        gen_apply_micro()
    for f in ("g15_rules_large.npz", "g16_garbage_large.npz", "g17_apply_micro.npz"):
        p = os.path.join(OUT, f)
        if os.path.exists(p):
            print(f"  {f}: {os.path.getsize(p) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
