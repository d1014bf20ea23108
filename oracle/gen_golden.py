#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, read-only, and oracle/_ref built by
`make -C oracle ref`).  Imports the reference's Python (src/, v1/python/) and its own CPU
`v0_core` (oracle/_ref) and records inputs + outputs as small .npz fixtures.  Nothing of the
reference's source text is stored -- only data (states, masks, visit counts, tensors).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Fixture groups:
  g1_rules.npz      reachable states from random src/ playouts: legal sets + every child transition
  g2_edges.npz      terminal states, the hand-built representative states, no-legal states
  g3_garbage.npz    synthetic random ("unreachable") states -> mask/metadata from reference v0_core
  g4_encode.npz     model-input planes and policy projection
  g5_tree_*.npz     variant-P tree searches with recorded evaluator outputs (+ src/mcts.py L check)
  g6_root_puct.npz  root bandit allocation
  g7_ops.npz        pack / finalize / self-play-step / trajectory-finalize op vectors
  g8_selfplay.npz   4-game root-PUCT self-play trace of the reference v1 runner (CPU)
  g11_loss.npz      training-loss values and head gradients from the reference's own loss functions
  g10_tree_selfplay.npz  full-tree self-play traces of the reference portable runner (subtree reuse on every move)
  g9_net.npz        network outputs for seeded weights (tiny / 6x64 / 10x128)
  g14_eval_arena.npz  the reference's evaluation arena (two tiny checkpoints, deterministic): outcome, moves, evaluations
  g13_legacy_waves.npz  root visit counts of src/mcts.py searches with batch_K = 16 / 4 (wave-batched leaves), incl. tree reuse
  g12_sparse_selfplay.npz  root-PUCT self-play traces of the reference v1 runner with sparse_ply = 2 and 3 (top-K lookahead)
"""
from __future__ import annotations

import os
import random
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("LZ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_ref"))
sys.path.insert(0, REF)
OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

import numpy as np
import torch

import v0_core  # the reference's own CPU build (oracle/_ref)
from src.game_state import GameState, Phase, Player
from src.move_generator import apply_move, generate_all_legal_moves
from src.neural_network import ChessNet, NUM_INPUT_CHANNELS, state_to_tensor
from src.policy_batch import TOTAL_DIM, action_to_index

torch.set_num_threads(4)

FIELDS = ("board", "marks_black", "marks_white", "phase", "current_player",
          "pending_marks_required", "pending_marks_remaining",
          "pending_captures_required", "pending_captures_remaining",
          "forced_removals_done", "move_count", "moves_since_capture")


def pack_states(states):
    n = len(states)
    out = {
        "board": np.zeros((n, 6, 6), np.int8),
        "marks_black": np.zeros((n, 6, 6), bool),
        "marks_white": np.zeros((n, 6, 6), bool),
    }
    for f in FIELDS[3:]:
        out[f] = np.zeros((n,), np.int64)
    for i, s in enumerate(states):
        out["board"][i] = np.array(s.board, np.int8)
        for (r, c) in s.marked_black:
            out["marks_black"][i, r, c] = True
        for (r, c) in s.marked_white:
            out["marks_white"][i, r, c] = True
        out["phase"][i] = s.phase.value
        out["current_player"][i] = s.current_player.value
        out["pending_marks_required"][i] = s.pending_marks_required
        out["pending_marks_remaining"][i] = s.pending_marks_remaining
        out["pending_captures_required"][i] = s.pending_captures_required
        out["pending_captures_remaining"][i] = s.pending_captures_remaining
        out["forced_removals_done"][i] = s.forced_removals_done
        out["move_count"][i] = s.move_count
        out["moves_since_capture"][i] = s.moves_since_capture
    return out


def to_torch(st):
    return [torch.from_numpy(st[f]) for f in FIELDS]


def prefixed(prefix, st):
    return {f"{prefix}_{k}": v for k, v in st.items()}


def legal_index_list(state):
    moves = generate_all_legal_moves(state)
    idx = [int(action_to_index(m, 6)) for m in moves]
    assert len(set(idx)) == len(idx)
    return moves, idx


# --------------------------------------------------------------------------------------------
# G1 / G2: rules
# --------------------------------------------------------------------------------------------
def representative_states():
    """The ten hand-built positions of the reference's cross-implementation test
    (tests/v1/test_portable_cpp_mcts.py:56-147), re-created here from their coordinates."""
    states = [GameState()]
    line = [[0] * 6 for _ in range(6)]
    for c in range(5):
        line[0][c] = 1
    for (r, c) in ((2, 0), (2, 2), (3, 4), (4, 1), (5, 3), (5, 5)):
        line[r][c] = -1
    s = GameState(board=line, phase=Phase.PLACEMENT, current_player=Player.BLACK)
    s = apply_move(s, {"phase": Phase.PLACEMENT, "action_type": "place", "position": (0, 5)}, quiet=True)
    states.append(s)
    s = apply_move(s, generate_all_legal_moves(s)[0], quiet=True)
    states.append(s)
    rem = [[0] * 6 for _ in range(6)]
    for (r, c) in ((0, 0), (0, 2), (1, 4), (3, 1), (5, 5)):
        rem[r][c] = 1
    for (r, c) in ((0, 5), (2, 1), (3, 4), (4, 0), (5, 2)):
        rem[r][c] = -1
    states.append(GameState(board=[x[:] for x in rem], phase=Phase.REMOVAL, current_player=Player.WHITE, marked_black={(0, 0)}))
    states.append(GameState(board=[x[:] for x in rem], phase=Phase.FORCED_REMOVAL, current_player=Player.WHITE, forced_removals_done=0))
    states.append(GameState(board=[x[:] for x in rem], phase=Phase.COUNTER_REMOVAL, current_player=Player.WHITE))
    mv = [[0] * 6 for _ in range(6)]
    for (r, c) in ((0, 0), (0, 1), (1, 0), (2, 1)):
        mv[r][c] = 1
    for (r, c) in ((2, 3), (2, 5), (3, 4), (4, 0), (5, 2)):
        mv[r][c] = -1
    movement = GameState(board=mv, phase=Phase.MOVEMENT, current_player=Player.BLACK, moves_since_capture=7)
    states.append(movement)
    states.append(apply_move(movement, {"phase": Phase.MOVEMENT, "action_type": "move",
                                        "from_position": (2, 1), "to_position": (1, 1)}, quiet=True))
    term = GameState(board=[x[:] for x in rem], phase=Phase.MOVEMENT, current_player=Player.BLACK)
    for (r, c) in ((2, 1), (3, 4)):
        term.board[r][c] = 0
    states.append(term)
    states.append(GameState(board=[x[:] for x in rem], phase=Phase.MOVEMENT, current_player=Player.WHITE,
                            move_count=GameState.MAX_MOVE_COUNT))
    # extra edge cases: stuck mover (no legal movement), no-legal MARK_SELECTION, full zhou column
    stuck = [[0] * 6 for _ in range(6)]
    stuck[0][0] = 1
    stuck[0][1] = -1; stuck[1][0] = -1
    for (r, c) in ((3, 3), (3, 4), (4, 3), (4, 4), (5, 0)):
        stuck[r][c] = -1
    for (r, c) in ((5, 5), (2, 5), (0, 5)):
        stuck[r][c] = 1
    states.append(GameState(board=stuck, phase=Phase.MOVEMENT, current_player=Player.BLACK))
    only = [[0] * 6 for _ in range(6)]
    only[0][0] = 1
    states.append(GameState(board=only, phase=Phase.MARK_SELECTION, current_player=Player.BLACK,
                            pending_marks_required=1, pending_marks_remaining=1))
    col = [[0] * 6 for _ in range(6)]
    for r in range(6):
        col[r][2] = -1
    for (r, c) in ((0, 0), (1, 1), (2, 3), (4, 4), (5, 5), (3, 0)):
        col[r][c] = 1
    states.append(GameState(board=col, phase=Phase.CAPTURE_SELECTION, current_player=Player.BLACK,
                            pending_captures_required=1, pending_captures_remaining=1))
    return states


def random_playout_states(num_games, seed, max_moves=400):
    rng = random.Random(seed)
    nonterm, term = [], []
    for _ in range(num_games):
        s = GameState()
        for _ in range(max_moves):
            if s.is_game_over():
                term.append(s)
                break
            moves = generate_all_legal_moves(s)
            if not moves:
                term.append(s)
                break
            nonterm.append(s)
            s = apply_move(s, rng.choice(moves), quiet=True)
    return nonterm, term


def gen_rules():
    nonterm, term = random_playout_states(260, 0x7777)
    rng = random.Random(1)
    by_phase = {}
    for s in nonterm:
        by_phase.setdefault(s.phase, []).append(s)
    print("[g1] reachable non-terminal states by phase:", {p.name: len(v) for p, v in by_phase.items()})
    quota = {Phase.PLACEMENT: 500, Phase.MARK_SELECTION: 350, Phase.REMOVAL: 120, Phase.MOVEMENT: 900,
             Phase.CAPTURE_SELECTION: 350, Phase.FORCED_REMOVAL: 120, Phase.COUNTER_REMOVAL: 120}
    chosen = []
    for p, lst in by_phase.items():
        rng.shuffle(lst)
        chosen.extend(lst[: quota.get(p, 100)])
    print("[g1] chosen:", len(chosen))
    st = pack_states(chosen)
    masks = np.zeros((len(chosen), TOTAL_DIM), bool)
    child_states, child_parent, child_action = [], [], []
    for i, s in enumerate(chosen):
        moves, idx = legal_index_list(s)
        order = np.argsort(idx)
        for k in order:
            masks[i, idx[k]] = True
            child_states.append(apply_move(s, moves[k], quiet=True))
            child_parent.append(i)
            child_action.append(idx[k])
    ch = pack_states(child_states)
    # sanity: the reference's own tensor ops agree with its Python on these states
    m_ref, meta_ref = v0_core.encode_actions_fast(*to_torch(st)[:10], 36, 144, 36, 4)
    assert np.array_equal(m_ref.numpy(), masks), "reference v0_core mask != reference python"
    codes = meta_ref.numpy()[np.array(child_parent), np.array(child_action)]
    applied = v0_core.batch_apply_moves(*to_torch(st), torch.from_numpy(codes.copy()),
                                        torch.tensor(child_parent, dtype=torch.int64))
    for f, t in zip(FIELDS, applied):
        assert np.array_equal(t.numpy().astype(ch[f].dtype), ch[f]), f"reference apply mismatch in {f}"
    np.savez_compressed(
        os.path.join(OUT, "g1_rules.npz"),
        legal_mask=np.packbits(masks, axis=1),
        metadata=meta_ref.numpy().astype(np.int8),
        child_parent=np.array(child_parent, np.int32),
        child_action=np.array(child_action, np.int16),
        **prefixed("s", st), **prefixed("c", ch),
    )
    print(f"[g1] states={len(chosen)} transitions={len(child_states)}")

    # ---- G2: terminals / representative / edge ----
    reps = representative_states()
    term = term[:150]
    edge = reps + term
    est = pack_states(edge)
    emask = np.zeros((len(edge), TOTAL_DIM), bool)
    status = np.zeros((len(edge),), np.int8)
    for i, s in enumerate(edge):
        _, idx = legal_index_list(s)
        emask[i, idx] = True
        w = s.get_winner()
        status[i] = (1 if w == Player.BLACK else -1) if w is not None else (2 if s.has_reached_move_limit() else 0)
    tmask, tmeta = v0_core.encode_actions_fast(*to_torch(est)[:10], 36, 144, 36, 4)
    np.savez_compressed(
        os.path.join(OUT, "g2_edges.npz"),
        py_legal_mask=np.packbits(emask, axis=1), status=status, num_representative=np.int64(len(reps)),
        tensor_mask=np.packbits(tmask.numpy(), axis=1), tensor_meta=tmeta.numpy().astype(np.int8),
        model_input=np.stack([state_to_tensor(s, s.current_player)[0].numpy() for s in edge]).astype(np.int8),
        **prefixed("s", est),
    )
    print(f"[g2] edge states={len(edge)} (representative={len(reps)}, terminal={len(term)})")
    return chosen


# --------------------------------------------------------------------------------------------
# G3: synthetic garbage (sampling recipe of tests/v0/cuda/test_fast_legal_mask_cuda.py:74-118)
# --------------------------------------------------------------------------------------------
def random_garbage(n, seed):
    g = torch.Generator().manual_seed(seed)
    board = torch.randint(-1, 2, (n, 6, 6), dtype=torch.int8, generator=g)
    mb = torch.randint(0, 2, (n, 6, 6), generator=g).to(torch.bool)
    mw = torch.randint(0, 2, (n, 6, 6), generator=g).to(torch.bool)
    phase = torch.randint(1, 8, (n,), dtype=torch.int64, generator=g)
    cur = torch.randint(0, 2, (n,), dtype=torch.int64, generator=g).mul(-2).add(1)
    z = torch.zeros(n, dtype=torch.int64)
    pmr = torch.randint(0, 3, (n,), dtype=torch.int64, generator=g)
    pcr = torch.randint(0, 3, (n,), dtype=torch.int64, generator=g)
    fr = torch.randint(0, 3, (n,), dtype=torch.int64, generator=g)
    mc = torch.randint(0, 150, (n,), dtype=torch.int64, generator=g)
    msc = torch.randint(0, 40, (n,), dtype=torch.int64, generator=g)
    return [board, mb, mw, phase, cur, z.clone(), pmr, z.clone(), pcr, fr, mc, msc]


def gen_garbage():
    n = 1500
    t = random_garbage(n, 0xF00DCAFE)
    # densify some boards so full rows / squares actually occur
    g = torch.Generator().manual_seed(5)
    dense = torch.where(torch.rand((n, 6, 6), generator=g) < 0.85, torch.ones((n, 6, 6), dtype=torch.int8),
                        -torch.ones((n, 6, 6), dtype=torch.int8))
    flip = (torch.rand((n, 1, 1), generator=g) < 0.5)
    dense = torch.where(flip, -dense, dense)
    use_dense = torch.rand((n, 1, 1), generator=g) < 0.4
    t[0] = torch.where(use_dense, dense, t[0]).to(torch.int8)
    sparse_marks = torch.rand((n, 1, 1), generator=g) < 0.5
    t[1] = torch.where(sparse_marks, t[1] & (torch.rand((n, 6, 6), generator=g) < 0.15), t[1])
    t[2] = torch.where(sparse_marks, t[2] & (torch.rand((n, 6, 6), generator=g) < 0.15), t[2])
    m1, meta1 = v0_core.encode_actions_fast(*t[:10], 36, 144, 36, 1)
    m4, meta4 = v0_core.encode_actions_fast(*t[:10], 36, 144, 36, 4)
    planes = v0_core.states_to_model_input(t[0], t[1], t[2], t[3], t[4])
    st = {f: x.numpy() for f, x in zip(FIELDS, t)}
    np.savez_compressed(
        os.path.join(OUT, "g3_garbage.npz"),
        mask_t217=np.packbits(m1.numpy(), axis=1), meta_t217=meta1.numpy().astype(np.int8),
        mask_t220=np.packbits(m4.numpy(), axis=1), meta_t220=meta4.numpy().astype(np.int8),
        model_input=planes.numpy().astype(np.int8),
        **prefixed("s", st),
    )
    print(f"[g3] garbage states={n}, legal per state avg={m4.sum(1).float().mean():.2f}")


# --------------------------------------------------------------------------------------------
# G4: projection
# --------------------------------------------------------------------------------------------
def gen_project(chosen):
    g = torch.Generator().manual_seed(11)
    n = 96
    sub = random.Random(3).sample(chosen, n)
    st = pack_states(sub)
    mask, _ = v0_core.encode_actions_fast(*to_torch(st)[:10], 36, 144, 36, 4)
    lp = [torch.log_softmax(torch.randn((n, 36), generator=g) * 2.0, dim=1) for _ in range(3)]
    # rows exercising the degenerate branches: no legal action, and all-(-inf) legal logits
    mask[0] = False
    lp[0][1] = float("-inf"); lp[1][1] = float("-inf"); lp[2][1] = float("-inf")
    probs, masked = v0_core.project_policy_logits_fast(lp[0], lp[1], lp[2], mask, 36, 144, 36, 4)
    np.savez_compressed(os.path.join(OUT, "g4_project.npz"), lp1=lp[0].numpy(), lp2=lp[1].numpy(),
                        lpmc=lp[2].numpy(), mask=np.packbits(mask.numpy(), axis=1),
                        probs=probs.numpy(), masked_logits=masked.numpy())
    print("[g4] projection rows:", n)


# --------------------------------------------------------------------------------------------
# G5: tree search (variant P) with a recording evaluator; cross-check vs src/mcts.py (variant L)
# --------------------------------------------------------------------------------------------
def small_model():
    torch.manual_seed(7)
    m = ChessNet(board_size=6, num_input_channels=NUM_INPUT_CHANNELS, trunk_channels=8, num_blocks=1,
                 policy_channels=4, value_channels=4, value_mlp_channels=8)
    m.eval()
    return m


def gen_tree(chosen):
    from v1.python.portable_mcts import PortableMCTS, PortableMCTSConfig, PortableTree
    from src.mcts import MCTS as LegacyMCTS

    model = small_model()
    rng = random.Random(9)
    pool = [s for s in representative_states() if not s.is_game_over() and generate_all_legal_moves(s)]
    by_phase = {}
    for s in chosen:
        by_phase.setdefault(s.phase, []).append(s)
    for p, lst in by_phase.items():
        pool.extend(rng.sample(lst, min(4, len(lst))))
    print("[g5] search roots:", len(pool))

    cases = []  # (root_idx, sims, noise)
    for i in range(len(pool)):
        cases.append((i, 8, False))
        cases.append((i, 64, False))
    for i in range(0, len(pool), 3):
        cases.append((i, 200, False))
        cases.append((i, 32, True))

    rec = {"eval_state": [], "eval_priors": [], "eval_value": [], "case_root": [], "case_sims": [],
           "case_noise_flag": [], "case_eval_start": [], "case_eval_count": [], "case_visits": [],
           "case_policy_t1": [], "case_policy_t01": [], "case_root_value": [], "case_chosen": [],
           "case_noise": [], "case_root_priors": []}

    for (ri, sims, noise_flag) in cases:
        cfg = PortableMCTSConfig(num_simulations=sims, exploration_weight=1.0, temperature=1.0,
                                 add_dirichlet_noise=noise_flag, sample_moves=False)
        search = PortableMCTS(model, cfg, "cpu")
        evals = []
        orig_eval = search.evaluate_states

        def recording_eval(states, _orig=orig_eval, _evals=evals):
            out = _orig(states)
            for row, s in enumerate(states):
                _evals.append((s.copy(), out.priors[row].numpy().copy(), float(out.values[row].item())))
            return out

        search.evaluate_states = recording_eval
        noise_rec = []
        orig_sample = torch.distributions.Dirichlet.sample

        def rec_sample(self, *a, **k):
            x = orig_sample(self, *a, **k)
            noise_rec.append(x.numpy().copy())
            return x

        torch.manual_seed(1234 + ri)
        torch.distributions.Dirichlet.sample = rec_sample
        try:
            out = search.search_batch([PortableTree(pool[ri])], temperatures=1.0)[0]
        finally:
            torch.distributions.Dirichlet.sample = orig_sample
        out01_policy = None
        # temperature 0.1 policy from the same visits (no new search: visits are deterministic)
        from v1.python.portable_mcts import policy_from_visits_and_priors
        idx = sorted(out.visit_counts)
        v = torch.tensor([out.visit_counts[i] for i in idx], dtype=torch.float32)
        p01 = policy_from_visits_and_priors(v, torch.ones_like(v), temperature=0.1)
        out01_policy = np.zeros(TOTAL_DIM, np.float32)
        out01_policy[idx] = p01.numpy()

        visits = np.zeros(TOTAL_DIM, np.int32)
        for a, cnt in out.visit_counts.items():
            visits[a] = cnt
        rec["case_root"].append(ri); rec["case_sims"].append(sims); rec["case_noise_flag"].append(noise_flag)
        rec["case_eval_start"].append(len(rec["eval_state"])); rec["case_eval_count"].append(len(evals))
        for (s, pri, val) in evals:
            rec["eval_state"].append(s); rec["eval_priors"].append(pri); rec["eval_value"].append(val)
        rec["case_visits"].append(visits)
        rec["case_policy_t1"].append(out.policy_dense.numpy()); rec["case_policy_t01"].append(out01_policy)
        rec["case_root_value"].append(out.root_value)
        rec["case_chosen"].append(-1 if out.chosen_action_index is None else out.chosen_action_index)
        nz = np.zeros(80, np.float32)
        if noise_rec:
            nz[: noise_rec[0].size] = noise_rec[0]
        rec["case_noise"].append(nz)
        rec["case_root_priors"].append(out.root_priors.numpy())

        # variant L cross-check (src/mcts.py, batch_K=1, no virtual loss, no noise)
        if not noise_flag and sims <= 64:
            legacy = LegacyMCTS(model, num_simulations=sims, exploration_weight=1.0, temperature=1.0,
                                device="cpu", add_dirichlet_noise=False, virtual_loss_weight=0.0, batch_K=1)
            legacy.search(pool[ri])
            lv = np.zeros(TOTAL_DIM, np.int32)
            for child in legacy.root.children:
                lv[int(action_to_index(child.move, 6))] = child.visit_count
            if not np.array_equal(lv, visits):
                print(f"   [g5] NOTE: src/mcts.py visits differ from portable at root {ri} sims {sims} "
                      f"(L1={np.abs(lv - visits).sum()})")
            rec.setdefault("legacy_case", []).append(len(rec["case_root"]) - 1)
            rec.setdefault("legacy_visits", []).append(lv)

    est = pack_states(rec["eval_state"])
    pst = pack_states(pool)
    pri = np.stack(rec["eval_priors"]).astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "g5_tree.npz"),
        eval_priors=pri, eval_value=np.array(rec["eval_value"], np.float32),
        case_root=np.array(rec["case_root"], np.int32), case_sims=np.array(rec["case_sims"], np.int32),
        case_noise_flag=np.array(rec["case_noise_flag"], bool),
        case_eval_start=np.array(rec["case_eval_start"], np.int64),
        case_eval_count=np.array(rec["case_eval_count"], np.int64),
        case_visits=np.stack(rec["case_visits"]), case_policy_t1=np.stack(rec["case_policy_t1"]),
        case_policy_t01=np.stack(rec["case_policy_t01"]),
        case_root_value=np.array(rec["case_root_value"], np.float64),
        case_chosen=np.array(rec["case_chosen"], np.int32), case_noise=np.stack(rec["case_noise"]),
        case_root_priors=np.stack(rec["case_root_priors"]),
        legacy_case=np.array(rec.get("legacy_case", []), np.int32),
        legacy_visits=np.stack(rec["legacy_visits"]) if rec.get("legacy_visits") else np.zeros((0, TOTAL_DIM), np.int32),
        **prefixed("e", est), **prefixed("r", pst),
    )
    print(f"[g5] cases={len(cases)} recorded evals={len(rec['eval_state'])}")


# --------------------------------------------------------------------------------------------
# G6: root bandit
# --------------------------------------------------------------------------------------------
def gen_root_puct():
    from v1.python.portable_root_puct import allocate_fixed_q_visits
    g = torch.Generator().manual_seed(21)
    R, A = 48, 40
    valid = torch.rand((R, A), generator=g) < 0.6
    valid[:, 0] = True
    valid[3] = False; valid[3, 5] = True  # single action
    pri = torch.rand((R, A), generator=g) * valid
    pri = pri / pri.sum(1, keepdim=True).clamp_min(1e-8)
    leaf = (torch.rand((R, A), generator=g) * 2 - 1) * valid
    leaf[7] = 0.25  # exact ties -> lowest index must win
    pri[7] = valid[7].float() / valid[7].sum()
    outs = {}
    for sims in (1, 16, 200, 1024):
        v, vs, rv = v0_core.root_puct_allocate_visits(pri, leaf, valid, sims, 1.0)
        outs[f"visits_{sims}"] = v.numpy(); outs[f"value_sum_{sims}"] = vs.numpy(); outs[f"root_{sims}"] = rv.numpy()
    v2, vs2, _ = v0_core.root_puct_allocate_visits(pri, leaf, valid, 64, 2.5)
    outs["visits_64_c25"] = v2.numpy(); outs["value_sum_64_c25"] = vs2.numpy()
    # spec twin agrees with the op (row 0..7)
    for r in range(8):
        a, b = allocate_fixed_q_visits(pri[r], leaf[r], valid[r], num_simulations=200, exploration_weight=1.0)
        assert torch.equal(a, torch.from_numpy(outs["visits_200"][r])), "spec twin mismatch"
    np.savez_compressed(os.path.join(OUT, "g6_root_puct.npz"), priors=pri.numpy(), leaf=leaf.numpy(),
                        valid=valid.numpy(), **outs)
    print("[g6] root puct rows:", R)


# --------------------------------------------------------------------------------------------
# G7: host ops
# --------------------------------------------------------------------------------------------
def gen_ops(chosen):
    rng = random.Random(17)
    sub = rng.sample(chosen, 61) + representative_states()[8:10] + [representative_states()[11]]
    st = pack_states(sub)
    t = to_torch(st)
    mask, meta = v0_core.encode_actions_fast(*t[:10], 36, 144, 36, 4)
    g = torch.Generator().manual_seed(31)
    probs = torch.rand((len(sub), 220), generator=g) * mask
    probs = probs / probs.sum(1, keepdim=True).clamp_min(1e-8)
    pack = v0_core.root_pack_sparse_actions(mask, probs, meta)
    names = ["terminal_mask", "valid_root_indices", "counts", "valid_mask", "legal_index_mat", "priors_mat",
             "action_code_mat", "pack_flat_idx", "action_codes_all", "parent_indices_all"]
    out = {f"pack_{n}": x.numpy() for n, x in zip(names, pack)}
    R, M = pack[3].shape
    visits = torch.floor(torch.rand((R, M), generator=g) * 20) * pack[3]
    visits[:, 0] += 1
    vsum = (torch.rand((R, M), generator=g) * 2 - 1) * visits
    temps = torch.where(torch.arange(R) % 2 == 0, torch.tensor(1.0), torch.tensor(0.1))
    fin = v0_core.root_finalize_from_visits(pack[4], pack[6], pack[3], visits, vsum, pack[1], len(sub), 220, temps, False)
    for n, x in zip(["policy_dense", "chosen_idx", "chosen_codes", "chosen_valid", "root_value"], fin):
        out[f"fin_{n}"] = x.numpy()
    out["fin_visits"] = visits.numpy(); out["fin_value_sum"] = vsum.numpy(); out["fin_temps"] = temps.numpy()

    # self_play_step_inplace on a copy of the batch
    t2 = [x.clone() for x in t]
    plies = torch.randint(0, 100, (len(sub),), generator=g)
    plies[5] = 95
    done = torch.zeros(len(sub), dtype=torch.bool)
    active = torch.arange(len(sub), dtype=torch.int64)
    active = active[active % 7 != 3]
    term_mask = pack[0][active]
    chosen_valid = fin[3][active].clone()
    chosen_codes = fin[2][active].clone()
    slots, res, soft = v0_core.self_play_step_inplace(*t2, plies, done, active, chosen_codes, term_mask,
                                                      chosen_valid, 96, 2.0)
    out["step_plies_in"] = np.array(plies.numpy()) - 0
    out["step_active"] = active.numpy(); out["step_codes"] = chosen_codes.numpy()
    out["step_term"] = term_mask.numpy(); out["step_valid"] = chosen_valid.numpy()
    out["step_slots"] = slots.numpy(); out["step_result"] = res.numpy(); out["step_soft"] = soft.numpy()
    out["step_plies_out"] = plies.numpy(); out["step_done_out"] = done.numpy()
    out.update({f"step_after_{f}": x.numpy() for f, x in zip(FIELDS, t2)})
    # (plies was mutated in place: recover the input)
    plies_in = plies.clone()
    valid_local = ~(term_mask | ~chosen_valid)
    plies_in[active[valid_local]] -= 1
    out["step_plies_in"] = plies_in.numpy()

    # finalize_trajectory_inplace
    G, Tm, S = 12, 20, 200
    vt = torch.full((S,), float("nan")); svt = torch.full((S,), float("nan"))
    signs = (torch.randint(0, 2, (S,), generator=g) * 2 - 1).to(torch.int8)
    sim = torch.full((G, Tm), -1, dtype=torch.int64)
    counts = torch.randint(0, Tm, (G,), generator=g)
    counts[2] = 0
    perm = torch.randperm(S, generator=g)
    k = 0
    for gi in range(G):
        c = int(counts[gi]); sim[gi, :c] = perm[k:k + c]; k += c
    fslots = torch.tensor([0, 2, 5, 7, 11], dtype=torch.int64)
    fres = torch.tensor([1.0, -1.0, 0.0, -1.0, 1.0]); fsoft = torch.tensor([0.5, -0.2, 0.0, -0.9, 0.3])
    r3 = v0_core.finalize_trajectory_inplace(vt, svt, signs, sim, counts, fslots, fres, fsoft)
    out.update(traj_signs=signs.numpy(), traj_step_index=sim.numpy(), traj_counts=counts.numpy(),
               traj_slots=fslots.numpy(), traj_result=fres.numpy(), traj_soft=fsoft.numpy(),
               traj_value_out=vt.numpy(), traj_soft_out=svt.numpy(),
               traj_final_slots=r3[0].numpy(), traj_final_counts=r3[1].numpy(), traj_counts_out=r3[2].numpy())
    np.savez_compressed(os.path.join(OUT, "g7_ops.npz"), probs=probs.numpy(),
                        mask=np.packbits(mask.numpy(), axis=1), meta=meta.numpy().astype(np.int8),
                        **prefixed("s", st), **out)
    print(f"[g7] ops batch={len(sub)} roots={R} maxA={M} finished={len(slots)}")


# --------------------------------------------------------------------------------------------
# G8: reference v1 root-PUCT self-play trace on CPU
# --------------------------------------------------------------------------------------------
def gen_selfplay():
    from v1.python.self_play_gpu_runner import self_play_v1_gpu
    model = small_model()
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    batch, stats = self_play_v1_gpu(
        model=model, num_games=4, mcts_simulations=32, temperature_init=1.0, temperature_final=0.1,
        temperature_threshold=10, exploration_weight=1.0, device="cpu", add_dirichlet_noise=False,
        soft_value_k=2.0, opening_random_moves=0, max_game_plies=512, sample_moves=False,
        concurrent_games=4)
    pol = batch.policy_targets.numpy()
    np.savez_compressed(
        os.path.join(OUT, "g8_selfplay.npz"),
        state_tensors=np.packbits(batch.state_tensors.numpy().astype(bool).reshape(batch.num_samples, -1), axis=1),
        legal_masks=np.packbits(batch.legal_masks.numpy(), axis=1),
        policy_targets=pol, value_targets=batch.value_targets.numpy(),
        soft_value_targets=batch.soft_value_targets.numpy(),
        num_positions=np.int64(stats.num_positions), black_wins=np.int64(stats.black_wins),
        white_wins=np.int64(stats.white_wins), draws=np.int64(stats.draws),
        avg_game_length=np.float64(stats.avg_game_length))
    print(f"[g8] self-play samples={batch.num_samples} W/L/D={stats.black_wins}/{stats.white_wins}/{stats.draws} "
          f"avg_len={stats.avg_game_length:.1f} ({stats.positions_per_sec:.1f} pos/s reference-on-CPU)")


# --------------------------------------------------------------------------------------------
# G13: the legacy search (src/mcts.py) with its wave-batched leaves: batch_K > 1, no virtual loss
# --------------------------------------------------------------------------------------------
def gen_legacy_waves():
    from src.mcts import MCTS as LegacyMCTS
    model = small_model()
    rng = random.Random(13)
    pool = [s for s in representative_states() if not s.is_game_over() and generate_all_legal_moves(s)]
    play = random_playout_states(6, seed=131, max_moves=90)[0]
    pool = rng.sample(pool, min(10, len(pool))) + rng.sample(play, min(14, len(play)))
    pool = [s for s in pool if not s.is_game_over() and generate_all_legal_moves(s)]
    roots, case_root, case_sims, case_k, case_moves, visits = [], [], [], [], [], []
    for ri, st in enumerate(pool):
        roots.append(st)
        for sims, k in ((16, 16), (40, 16), (200, 16), (50, 4)):
            if sims == 200 and ri % 3:
                continue
            mcts = LegacyMCTS(model, num_simulations=sims, exploration_weight=1.0, temperature=1.0, device="cpu",
                              add_dirichlet_noise=False, virtual_loss_weight=0.0, batch_K=k)
            # three consecutive moves with tree reuse (advance_root): most-visited child, lowest index on ties
            cur = st.copy()
            rows, picks = [], []
            for _ in range(3):
                if cur.is_game_over() or not generate_all_legal_moves(cur):
                    break
                mcts.search(cur)
                v = np.zeros(TOTAL_DIM, np.int32)
                for child in mcts.root.children:
                    v[int(action_to_index(child.move, 6))] = child.visit_count
                rows.append(v)
                best = int(np.flatnonzero(v == v.max())[0])
                mv = next(ch.move for ch in mcts.root.children if int(action_to_index(ch.move, 6)) == best)
                picks.append(best)
                cur = apply_move(cur.copy(), mv, quiet=True)
                mcts.advance_root(mv)
            case_root.append(ri); case_sims.append(sims); case_k.append(k)
            case_moves.append(len(rows))
            while len(rows) < 3:
                rows.append(np.zeros(TOTAL_DIM, np.int32))
            visits.append(np.stack(rows))
    np.savez_compressed(
        os.path.join(OUT, "g13_legacy_waves.npz"),
        case_root=np.array(case_root, np.int32), case_sims=np.array(case_sims, np.int32),
        case_k=np.array(case_k, np.int32), case_moves=np.array(case_moves, np.int32), case_visits=np.stack(visits),
        **prefixed("r", pack_states(roots)))
    print(f"[g13] legacy wave searches: roots={len(roots)} cases={len(case_root)}")


# --------------------------------------------------------------------------------------------
# G12: reference v1 runner with the experimental multi-ply lookahead (sparse_ply > 1) on CPU
# --------------------------------------------------------------------------------------------
def gen_sparse_selfplay():
    from v1.python.self_play_gpu_runner import self_play_v1_gpu
    out = {}
    for tag, ply, top_k in (("p2k4", 2, 4), ("p3k3", 3, 3)):
        model = small_model()
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        batch, stats = self_play_v1_gpu(
            model=model, num_games=3, mcts_simulations=16, temperature_init=1.0, temperature_final=0.1,
            temperature_threshold=10, exploration_weight=1.0, device="cpu", add_dirichlet_noise=False,
            soft_value_k=2.0, opening_random_moves=0, max_game_plies=48, sample_moves=False,
            concurrent_games=3, sparse_ply=ply, sparse_top_k=top_k)
        n = batch.num_samples
        out[f"{tag}_state_tensors"] = np.packbits(batch.state_tensors.numpy().astype(bool).reshape(n, -1), axis=1)
        out[f"{tag}_legal_masks"] = np.packbits(batch.legal_masks.numpy(), axis=1)
        out[f"{tag}_policy_targets"] = batch.policy_targets.numpy()
        out[f"{tag}_value_targets"] = batch.value_targets.numpy()
        out[f"{tag}_soft_value_targets"] = batch.soft_value_targets.numpy()
        out[f"{tag}_config"] = np.asarray([ply, top_k, 3, 16, 48], np.int64)
        out[f"{tag}_outcome"] = np.asarray([stats.black_wins, stats.white_wins, stats.draws], np.int64)
        print(f"[g12/{tag}] samples={n} W/L/D={stats.black_wins}/{stats.white_wins}/{stats.draws}")
    np.savez_compressed(os.path.join(OUT, "g12_sparse_selfplay.npz"), **out)


# --------------------------------------------------------------------------------------------
# G10: reference portable (full tree, subtree reuse) self-play traces on CPU
# --------------------------------------------------------------------------------------------
def gen_tree_selfplay():
    from v1.python.portable_self_play import self_play_v1_portable
    out = {}
    for tag, kw in (
        ("a", dict(num_games=3, mcts_simulations=24, max_game_plies=48, concurrent_games=3)),
        ("b", dict(num_games=2, mcts_simulations=16, max_game_plies=14, concurrent_games=2,
                   policy_target_temperature=1.0, policy_target_prior_pseudocount=0.5)),
    ):
        model = small_model()
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        # Every network evaluation of the run (planes -> priors over the 220 actions, value), recorded at the reference's
        # own hand-off (PortableMCTS.evaluate_states).  The tiny random net's priors are nearly uniform and the games
        # turn on their last bits (a 1-ulp change alters the trace), and a CPU with another vector ISA rounds the
        # convolutions differently -- a parity test on another host replays this table instead of re-evaluating.
        from v1.python import portable_mcts as _pm
        rec = {}
        orig = _pm.PortableMCTS.evaluate_states

        def recording(self, states, _orig=orig, _rec=rec):
            ev = _orig(self, states)
            planes = np.packbits(ev.model_inputs.numpy().astype(bool).reshape(len(states), -1), axis=1)
            for i in range(len(states)):
                key = planes[i].tobytes()
                val = (ev.priors[i].numpy().copy(), float(ev.values[i]))
                if key in _rec:
                    assert np.array_equal(_rec[key][0], val[0]) and _rec[key][1] == val[1], "evaluation is not batch-invariant"
                _rec[key] = val
            return ev

        _pm.PortableMCTS.evaluate_states = recording
        try:
            batch, stats = self_play_v1_portable(
                model=model, temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
                exploration_weight=1.0, device="cpu", add_dirichlet_noise=False, soft_value_k=2.0,
                opening_random_moves=0, sample_moves=False, **kw)
        finally:
            _pm.PortableMCTS.evaluate_states = orig
        keys = sorted(rec)
        out[f"{tag}_eval_planes"] = np.frombuffer(b"".join(keys), np.uint8).reshape(len(keys), -1).copy()
        out[f"{tag}_eval_priors"] = np.stack([rec[k][0] for k in keys]).astype(np.float32)
        out[f"{tag}_eval_values"] = np.asarray([rec[k][1] for k in keys], np.float32)
        print(f"[g10/{tag}] {len(keys)} distinct evaluated positions recorded")
        n = batch.num_samples
        out.update({
            f"{tag}_state_tensors": np.packbits(batch.state_tensors.numpy().astype(bool).reshape(n, -1), axis=1),
            f"{tag}_legal_masks": np.packbits(batch.legal_masks.numpy(), axis=1),
            f"{tag}_policy_targets": batch.policy_targets.numpy(),
            f"{tag}_value_targets": batch.value_targets.numpy(),
            f"{tag}_soft_value_targets": batch.soft_value_targets.numpy(),
            f"{tag}_outcome": np.array([stats.black_wins, stats.white_wins, stats.draws], np.int64),
            f"{tag}_config": np.array([kw["num_games"], kw["mcts_simulations"], kw["max_game_plies"]], np.int64),
        })
        print(f"[g10/{tag}] tree self-play samples={n} avg_len={stats.avg_game_length:.1f} "
              f"({stats.positions_per_sec:.1f} pos/s reference-on-CPU)")
    np.savez_compressed(os.path.join(OUT, "g10_tree_selfplay.npz"), **out)


# --------------------------------------------------------------------------------------------
# G14: the reference's evaluation arena (scripts/eval_checkpoint.py::_eval_worker_v1, backend v1) on CPU
# --------------------------------------------------------------------------------------------
def _fnv64(rows: np.ndarray) -> np.ndarray:
    """FNV-1a over the bytes of each row (uint8[N, K]) -> uint64[N]: key of a packed input-plane row."""
    h = np.full(rows.shape[0], 0xCBF29CE484222325, np.uint64)
    for j in range(rows.shape[1]):
        h = (h ^ rows[:, j].astype(np.uint64)) * np.uint64(0x100000001B3)
    return h


def gen_eval_arena():
    """Two tiny checkpoints play `num_games` games against each other through the reference's own arena worker
    (deterministic picks, no random openings): the outcome tuple, every game's move sequence, and every network
    evaluation of both agents (so that a run on another host replays them instead of re-rounding the convolutions;
    value for every evaluated position, the three head rows for the positions that were searched as roots)."""
    import importlib
    ec = importlib.import_module("scripts.eval_checkpoint")
    from src.neural_network import bucket_logits_to_scalar
    G, SIMS = 4, 16
    models = {}
    for tag, seed in (("chall", 7), ("opp", 8)):
        torch.manual_seed(seed)
        m = ChessNet(board_size=6, num_input_channels=NUM_INPUT_CHANNELS, trunk_channels=8, num_blocks=1,
                     policy_channels=4, value_channels=4, value_mlp_channels=8).eval()
        models[tag] = m
    rec = {"chall": {}, "opp": {}}

    class Recording(torch.nn.Module):
        def __init__(self, tag):
            super().__init__()
            self.tag, self.inner = tag, models[tag]

        def forward(self, x):
            out = self.inner(x)
            keys = _fnv64(np.packbits(x.detach().cpu().numpy().astype(bool).reshape(x.shape[0], -1), axis=1))
            val = bucket_logits_to_scalar(out[3].float(), num_bins=int(out[3].shape[1])).detach().cpu().numpy()
            heads = torch.cat([o.reshape(x.shape[0], -1) for o in out[:3]], dim=1).detach().cpu().numpy().astype(np.float32)
            for i, k in enumerate(keys.tolist()):
                v = (heads[i].copy(), np.float32(val[i]))
                if k in rec[self.tag]:
                    assert np.array_equal(rec[self.tag][k][0], v[0]) and rec[self.tag][k][1] == v[1], "not batch-invariant"
                rec[self.tag][k] = v
            return out

    ec._load_model_from_checkpoint = lambda path, device: Recording("chall" if "chall" in str(path) else "opp").eval()
    moves, roots, ids = {}, set(), {}
    real_apply = ec.apply_move

    def logging_apply(state, move, quiet=True):
        g = ids.pop(id(state), None)
        if g is None:
            g = len(moves)
            moves[g] = []
        moves[g].append(int(action_to_index(move, 6)))
        planes = state_to_tensor(state, state.current_player).numpy().astype(bool).reshape(1, -1)
        roots.add(int(_fnv64(np.packbits(planes, axis=1))[0]))
        nxt = real_apply(state, move, quiet=quiet)
        ids[id(nxt)] = g
        logging_apply.keep.append(nxt)                         # keep ids unique while the game is alive
        return nxt
    logging_apply.keep = []
    ec.apply_move = logging_apply
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    try:
        result = ec._eval_worker_v1(0, list(range(G)), G, "cpu", SIMS, 0.1, "chall.pt", "opp.pt", 0, G, 0, False, "v1",
                                    "python", 1)
    finally:
        ec.apply_move = real_apply
    L = max(len(v) for v in moves.values())
    seq = np.full((G, L), -1, np.int32)
    for g, v in moves.items():
        seq[g, :len(v)] = v
    out = {"result": np.asarray(result, np.int64), "moves": seq, "config": np.asarray([G, SIMS], np.int64)}
    for tag in ("chall", "opp"):
        keys = np.asarray(sorted(rec[tag]), np.uint64)
        assert len(set(keys.tolist())) == len(keys)
        out[f"{tag}_keys"] = keys
        out[f"{tag}_values"] = np.asarray([rec[tag][int(k)][1] for k in keys], np.float32)
        rk = np.asarray([k for k in keys.tolist() if k in roots], np.uint64)
        out[f"{tag}_root_keys"] = rk
        out[f"{tag}_root_heads"] = np.stack([rec[tag][int(k)][0] for k in rk]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g14_eval_arena.npz"), **out)
    print(f"[g14] result={result} game lengths={[len(v) for v in moves.values()]} "
          f"evaluations chall/opp={len(rec['chall'])}/{len(rec['opp'])}")


# --------------------------------------------------------------------------------------------
# G11: training loss (values + autograd gradients of the reference's functions)
# --------------------------------------------------------------------------------------------
def gen_loss():
    from src.policy_batch import build_combined_logits, masked_log_softmax, batched_policy_loss
    from src.neural_network import scalar_to_bucket_twohot, scalar_to_wdl
    from v1.python.train_bridge import _bucket_logits_to_wdl_probs
    g = torch.Generator().manual_seed(41)
    B = 48
    raw1 = torch.randn((B, 36), generator=g); raw2 = torch.randn((B, 36), generator=g); raw3 = torch.randn((B, 36), generator=g)
    vlog = torch.randn((B, 101), generator=g) * 2
    mask = torch.rand((B, 220), generator=g) < 0.12
    comb_valid = torch.isfinite(build_combined_logits(raw1, raw2, raw3, 6))
    mask &= comb_valid
    mask[:, 0] = True
    mask[5] = False                       # a row without any legal action
    mask[6] = False; mask[6, 216] = True  # only the auxiliary action
    target = torch.rand((B, 220), generator=g) * mask
    target = target / target.sum(1, keepdim=True).clamp_min(1e-8)
    target[7] = 0                          # empty target row
    value = torch.randint(-1, 2, (B,), generator=g).float()
    value[8] = 0.37; value[9] = -0.81      # non-integer values (soft-value mixing inputs)
    soft = torch.rand((B,), generator=g) * 2 - 1
    out = dict(raw1=raw1.numpy(), raw2=raw2.numpy(), raw3=raw3.numpy(), value_logits=vlog.numpy(),
               mask=mask.numpy(), target=target.numpy(), value=value.numpy(), soft=soft.numpy())
    for tag, alpha, anti, dw in (("a", 0.0, 0.0, 1.0), ("b", 0.35, -0.15, 0.4)):
        l1 = torch.log_softmax(raw1, 1).requires_grad_(True); l2 = torch.log_softmax(raw2, 1).requires_grad_(True)
        l3 = torch.log_softmax(raw3, 1).requires_grad_(True); vl = vlog.clone().requires_grad_(True)
        raw_v = value.view(-1, 1); bv = raw_v
        draw = raw_v.abs().lt(1e-8)
        if abs(anti) > 1e-9:
            bv = bv.clone(); bv[draw] = anti
        mixed = torch.clamp((1.0 - alpha) * bv + alpha * soft.view(-1, 1), -1.0, 1.0)
        tgt = scalar_to_bucket_twohot(mixed, num_bins=101).float()
        vlp = torch.log_softmax(vl.float(), dim=-1)
        bucket = -(tgt * vlp).sum(-1).mean()
        wdl_aux = -(scalar_to_wdl(raw_v).float() * torch.log(_bucket_logits_to_wdl_probs(vl).clamp_min(1e-8))).sum(-1)
        comb = build_combined_logits(l1, l2, l3, board_size=6)
        logp = masked_log_softmax(comb, mask, dim=1)
        pol = batched_policy_loss(logp, target, mask, raw_v, dw)
        loss = pol + bucket
        loss.backward()
        out.update({f"{tag}_params": np.array([alpha, anti, dw], np.float32), f"{tag}_loss": loss.detach().numpy(),
                    f"{tag}_policy_loss": pol.detach().numpy(), f"{tag}_bucket_loss": bucket.detach().numpy(),
                    f"{tag}_wdl_aux": wdl_aux.detach().numpy(), f"{tag}_g1": l1.grad.numpy(), f"{tag}_g2": l2.grad.numpy(),
                    f"{tag}_g3": l3.grad.numpy(), f"{tag}_gv": vl.grad.numpy()})
        print(f"[g11/{tag}] loss={float(loss):.6f} policy={float(pol):.6f} bucket={float(bucket):.6f}")
    np.savez_compressed(os.path.join(OUT, "g11_loss.npz"), **out)


# --------------------------------------------------------------------------------------------
# G9: network
# --------------------------------------------------------------------------------------------
def gen_net(chosen):
    sub = random.Random(23).sample(chosen, 8)
    x = torch.cat([state_to_tensor(s, s.current_player) for s in sub], dim=0)
    out = {"inputs": x.numpy().astype(np.int8)}
    keys = {}
    for name, kw, seed in (
        ("tiny", dict(trunk_channels=8, num_blocks=1, policy_channels=4, value_channels=4, value_mlp_channels=8), 7),
        ("b6c64", dict(trunk_channels=64, num_blocks=6), 20260314),
        ("b10c128", dict(), 20260314),
    ):
        torch.manual_seed(seed)
        m = ChessNet(board_size=6, num_input_channels=NUM_INPUT_CHANNELS, **kw)
        # make BatchNorm statistics non-trivial so folding is actually exercised
        g = torch.Generator().manual_seed(seed + 1)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) * 0.5 + 0.75)
                mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.75)
                mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
        m.eval()
        with torch.inference_mode():
            lp1, lp2, lpm, v = m(x)
        out[f"{name}_lp1"] = lp1.numpy(); out[f"{name}_lp2"] = lp2.numpy()
        out[f"{name}_lpmc"] = lpm.numpy(); out[f"{name}_value_logits"] = v.numpy()
        sd = m.state_dict()
        keys[name] = [(k, tuple(t.shape)) for k, t in sd.items()]
        out[f"{name}_num_params"] = np.int64(sum(p.numel() for p in m.parameters()))
    import json
    with open(os.path.join(OUT, "g9_net_keys.json"), "w") as f:
        json.dump({k: [[n, list(s)] for n, s in v] for k, v in keys.items()}, f)
    np.savez_compressed(os.path.join(OUT, "g9_net.npz"), **out)
    print("[g9] nets:", {k: int(out[f"{k}_num_params"]) for k in keys})


def main():
    which = set(sys.argv[1:])
    chosen = gen_rules() if (not which or which & {"g1", "g4", "g5", "g7", "g9"}) else None
    if not which or "g3" in which:
        gen_garbage()
    if not which or "g4" in which:
        gen_project(chosen)
    if not which or "g5" in which:
        gen_tree(chosen)
    if not which or "g6" in which:
        gen_root_puct()
    if not which or "g7" in which:
        gen_ops(chosen)
    if not which or "g8" in which:
        gen_selfplay()
    if not which or "g9" in which:
        gen_net(chosen)
    if not which or "g10" in which:
        gen_tree_selfplay()
    if not which or "g14" in which:
        gen_eval_arena()
    if not which or "g13" in which:
        gen_legacy_waves()
    if not which or "g12" in which:
        gen_sparse_selfplay()
    if not which or "g11" in which:
        gen_loss()
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
