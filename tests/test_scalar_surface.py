"""CPU: the scalar half of the `v0_core` surface (liuzhou_amd/v0_scalar.py over include/liuzhou_scalar.h in the host library)
against the reference's own compiled module: the committed record of its behaviour (tests/golden/g18_scalar_surface.npz,
written by oracle/gen_golden_scalar.py from oracle/_ref), the live module when it is there, and the reference's
`v0/python/move_generator.py` running unmodified over the drop-in."""
import glob
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from liuzhou_amd import _lib, v0_core
from tests.golden_utils import load
from tests.scalar_walk import record, state_row

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LZ_REFERENCE", "/root/reference")

# what v0/src/bindings/module.cpp:877-1155 binds next to the tensor operators (names as data)
SURFACE = ("Phase Player ActionType MoveRecord ActionCode GameState TensorStateBatch generate_placement_positions "
           "apply_placement_move generate_mark_targets apply_mark_selection process_phase2_removals generate_movement_moves "
           "has_legal_movement_moves apply_movement_move generate_capture_targets apply_capture_selection apply_forced_removal "
           "handle_no_moves_phase3 apply_counter_removal_phase3 generate_legal_moves_phase1 apply_move_phase1 "
           "generate_legal_moves_phase3 has_legal_moves_phase3 apply_move_phase3 generate_all_legal_moves_struct "
           "generate_moves_with_codes generate_forced_removal_moves_struct generate_no_moves_options_struct "
           "generate_counter_removal_moves_struct encode_action_codes encode_action_code apply_move_struct "
           "tensor_batch_from_game_states tensor_batch_to_game_states").split()


def _reference_module():
    found = glob.glob(os.path.join(ROOT, "oracle", "_ref", "v0_core*.so"))
    if not found:
        return None
    spec = importlib.util.spec_from_file_location("v0_core", found[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _same(got, want):
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape, k
        if not np.array_equal(got[k], want[k]):
            rows = np.nonzero((got[k] != want[k]).reshape(len(want[k]), -1).any(1))[0]
            raise AssertionError(f"{k}: {len(rows)} rows differ, first {rows[:5].tolist()}")


def test_host_library_exports_the_scalar_abi():
    header = open(os.path.join(ROOT, "include", "liuzhou_scalar.h")).read()
    declared = set(re.findall(r"LZ_API\s+int\s+(lz_scalar_\w+)\s*\(", header))
    assert len(declared) == 6
    H = _lib.host_lib()
    for sym in declared:
        assert hasattr(H, sym), f"{sym} declared in liuzhou_scalar.h but not exported by libliuzhou_host.so"


def test_walk_equals_the_reference_record():
    """1 269 states (5 random games to the end, crafted forced-removal / stuck-player starts, 250 unreachable states),
    every generated list, 37 917 probe calls (91 % of them raise in the reference; raised-or-not and the returned state
    must agree), action codes, the tensor-batch round trip: byte-identical to the reference module's record."""
    want = load("g18_scalar_surface.npz")
    assert want["states"].shape[0] > 1200 and set(np.unique(want["states"][:, 108]).tolist()) == {1, 2, 3, 4, 5, 6, 7}
    _same(record(v0_core), {k: want[k] for k in want.files})


def test_walk_equals_the_live_reference_module_on_another_seed():
    R = _reference_module()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    _same(record(v0_core, seed=77, games=3, garbage=400), record(R, seed=77, games=3, garbage=400))
    # pybind's export_values(): later enums overwrite earlier names
    for name in ("PLACEMENT", "BLACK", "WHITE", "PLACE", "FORCED_REMOVAL", "COUNTER_REMOVAL", "PROCESS_REMOVAL"):
        ours, ref = getattr(v0_core, name), getattr(R, name)
        assert type(ours).__name__ == type(ref).__name__ and int(ours) == int(ref), name


def test_surface_names_types_and_errors():
    spec = importlib.util.spec_from_file_location("v0_core_dropin_check", os.path.join(ROOT, "liuzhou_amd", "dropin", "v0_core.py"))
    dropin_check = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dropin_check)                            # what `import v0_core` gives with the drop-in directory on the path
    for name in SURFACE:
        assert hasattr(v0_core, name), name
        assert getattr(dropin_check, name) is getattr(v0_core, name), name
    core = v0_core
    assert core.Phase.MOVEMENT.value == 4 and core.Player.WHITE.value == -1 and core.ActionType.NO_MOVES_REMOVAL.value == 7
    with pytest.raises(TypeError):
        core.MoveRecord()                                           # the reference binds no constructor
    m = core.MoveRecord.movement((1, 2), (1, 3))
    assert (m.phase, m.action_type, m.action_type_name) == (core.Phase.MOVEMENT, core.ActionType.MOVE, "move")
    assert m.position is None and m.from_position == (1, 2) and m.to_position == (1, 3)
    assert m.to_dict() == {"phase": core.Phase.MOVEMENT, "action_type": "move", "from_position": (1, 2), "to_position": (1, 3)}
    p = core.MoveRecord.process_removal()
    assert p.position is None and p.to_dict() == {"phase": core.Phase.REMOVAL, "action_type": "process_removal"}
    assert core.encode_action_code(p).to_tuple() == (8, 0, 0, 0) and core.encode_action_code(m).to_tuple() == (2, 8, 9, 0)
    c = core.ActionCode()
    c.kind, c.primary = 3, 17
    assert c.to_tuple() == (3, 17, 0, 0)
    g = core.GameState()
    assert g.phase == core.Phase.PLACEMENT and g.current_player == core.Player.BLACK and g.board == [[0] * 6] * 6
    assert g.marked_black == [] and not g.is_board_full() and g.count_player_pieces(core.Player.BLACK) == 0
    with pytest.raises(RuntimeError):
        g.board = [[0] * 6] * 5
    with pytest.raises(RuntimeError):
        g.board = [[2] + [0] * 5] + [[0] * 6] * 5
    with pytest.raises(RuntimeError):
        g.marked_white = [(6, 0)]
    g.marked_white = [(3, 3), (0, 1)]
    assert g.marked_white == [(0, 1), (3, 3)]                        # ascending cells, like MarkSet::ToVector
    h = g.copy()
    h.switch_player(); h.move_count = 9
    assert g.current_player == core.Player.BLACK and g.move_count == 0 and h.current_player == core.Player.WHITE
    nxt = core.apply_placement_move(g, (2, 2))
    assert nxt.board[2][2] == 1 and g.board[2][2] == 0 and nxt.get_player_pieces(core.Player.BLACK) == [(2, 2)]
    with pytest.raises(RuntimeError):
        core.apply_placement_move(nxt, (2, 2))
    with pytest.raises(RuntimeError):
        core.has_legal_movement_moves(g)
    with pytest.raises(RuntimeError):
        core.tensor_batch_from_game_states([])
    with pytest.raises(RuntimeError, match="phase"):
        core.apply_move_struct(g, core.MoveRecord.mark((0, 0)))
    # any object with the attributes of src.game_state.GameState is accepted as a state
    class Like:
        board = nxt.board; phase = core.Phase.PLACEMENT; current_player = core.Player.WHITE
        marked_black = set(); marked_white = set(); move_count = 1
    assert len(core.generate_placement_positions(Like())) == 35
    b = core.tensor_batch_from_game_states([g, nxt])
    assert b.board.dtype == torch.int8 and b.marks_black.dtype == torch.bool and b.phase.dtype == torch.int64
    assert b.board_size == 6 and b.device() == torch.device("cpu") and bool(b.mask_alive.all())
    assert not hasattr(b, "moves_since_capture")                     # carried inside, not bound (module.cpp:1112-1144)
    back = core.tensor_batch_to_game_states(b.clone())
    assert [state_row(s) for s in back] == [state_row(g), state_row(nxt)]


CHILD = r'''
import random, sys
import torch
import v0_core
assert "liuzhou_amd" in v0_core.__file__, v0_core.__file__
from v0.python.move_generator import generate_all_legal_moves, generate_all_legal_moves_with_codes, apply_move
from v0.python.move_generator import _generate_moves_forced_removal, _generate_moves_no_moves, _generate_moves_counter_removal
from src.move_generator import generate_all_legal_moves as py_generate, apply_move as py_apply
from src.move_generator import _generate_moves_forced_removal as py_forced, _generate_moves_no_moves as py_no_moves
from src.move_generator import _generate_moves_counter_removal as py_counter
from src.game_state import GameState, Phase
FIELDS = ("board", "phase", "current_player", "marked_black", "marked_white", "forced_removals_done", "move_count",
          "pending_marks_required", "pending_marks_remaining", "pending_captures_required", "pending_captures_remaining")
random.seed(11)
plies = 0
for game in range(4):
    s = GameState()
    while True:
        want = py_generate(s)
        if not want:                 # (the C++-backed module cannot see the no-capture draw: its GameState binding has
            break                    #  no moves_since_capture -- the reference's own limitation, module.cpp:974-1006)
        got, codes = generate_all_legal_moves_with_codes(s)
        assert got == want and generate_all_legal_moves(s) == want, (got[:2], want[:2])
        assert codes.dtype == torch.int32 and tuple(codes.shape) == (len(want), 4)
        if s.phase == Phase.FORCED_REMOVAL:
            assert _generate_moves_forced_removal(s) == py_forced(s)
        if s.phase == Phase.COUNTER_REMOVAL:
            assert _generate_moves_counter_removal(s) == py_counter(s)
        if s.phase == Phase.MOVEMENT and want[0]["action_type"] == "no_moves_remove":
            assert _generate_moves_no_moves(s) == py_no_moves(s)
        for m in random.sample(want, min(3, len(want))):
            a, b = apply_move(s, m, quiet=True), py_apply(s, m, quiet=True)
            for f in FIELDS:
                assert getattr(a, f) == getattr(b, f), (f, m)
        core_state = v0_core.tensor_batch_to_game_states(v0_core.tensor_batch_from_game_states([apply_move(v0_core.GameState(), {"phase": Phase.PLACEMENT, "action_type": "place", "position": (0, 0)})]))[0]
        assert isinstance(apply_move(core_state, {"phase": Phase.PLACEMENT, "action_type": "place", "position": (1, 1)}), v0_core.GameState)
        s = py_apply(s, random.choice(want), quiet=True)
        plies += 1
print("plies", plies)
'''


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "v0", "python", "move_generator.py")), reason="reference not mounted")
def test_reference_move_generator_module_runs_unmodified_over_the_drop_in(tmp_path):
    """`v0/python/move_generator.py` is the reference's C++-backed stand-in for `src.move_generator`; with `import v0_core`
    resolving to the drop-in it must agree with the pure-Python engine on every state of four random games (moves, their
    order, the action-code tensor, the successor states), without an edit."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "liuzhou_amd", "dropin"), ROOT, REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert int(r.stdout.split()[-1]) > 300


def test_status_and_shape_entry_points_against_the_oracle_and_a_direct_restatement():
    """`lz_scalar_status` (GetWinner / IsGameOver, game_state.cpp:58-79) against the oracle's game status on the golden
    states, and `lz_scalar_piece_in_shape` (rule_engine.cpp:194-208) against a cell-by-cell restatement of the shape
    rules (2x2 block of own unmarked pieces; a full row / column counting the probed cell whatever its mark) on random
    boards with marks."""
    import ctypes as C
    from liuzhou_amd import v0_scalar as V
    from oracle import lz_oracle as O
    from tests.golden_utils import states as golden_states
    fn = V._host()
    st = golden_states(load("g1_rules.npz"), "s")
    n = int(st["board"].shape[0])
    rng = np.random.default_rng(9)
    for i in rng.integers(0, n, 400).tolist():
        cs = O.state_from_batch(st, i)
        c = V._CState()
        board = np.asarray(st["board"][i]).reshape(36)
        for j in range(36):
            c.board[j] = int(board[j])
        c.phase, c.current_player = int(st["phase"][i]), int(st["current_player"][i])
        c.move_count, c.moves_since_capture = int(st["move_count"][i]), int(st["moves_since_capture"][i])
        w, over = C.c_int32(9), C.c_int32(9)
        assert fn["lz_scalar_status"](C.byref(c), C.byref(w), C.byref(over)) == 0
        g = O.game_status(cs)                                        # 0 running, +1 / -1 winner, 2 draw
        assert over.value == (1 if g != 0 else 0) and w.value == (g if g in (1, -1) else 0)
    for trial in range(300):
        board = rng.choice([-1, 0, 1], size=(6, 6), p=[0.45, 0.1, 0.45])
        if trial % 5 == 0:
            board[rng.integers(0, 6), :] = 1                         # a full row now and then
        if trial % 7 == 0:
            board[:, rng.integers(0, 6)] = -1
        marks = {1: rng.random((6, 6)) < 0.15, -1: rng.random((6, 6)) < 0.15}
        c = V._CState()
        for j in range(36):
            c.board[j] = int(board.reshape(36)[j])
            c.marks_black[j], c.marks_white[j] = int(marks[1].reshape(36)[j]), int(marks[-1].reshape(36)[j])
        for use_marks in (0, 1):
            for player in (1, -1):
                m = marks[player] if use_marks else np.zeros((6, 6), dtype=bool)
                free = (board == player) & ~m
                for r in range(6):
                    for col in range(6):
                        want = 0
                        if board[r, col] == player:
                            sq = any(0 <= rr < 5 and 0 <= cc < 5 and free[rr:rr + 2, cc:cc + 2].all()
                                     for rr in (r, r - 1) for cc in (col, col - 1))
                            row = all(free[r, k] or k == col for k in range(6))
                            line = all(free[k, col] or k == r for k in range(6))
                            want = int(sq or row or line)
                        got = C.c_int32(7)
                        assert fn["lz_scalar_piece_in_shape"](C.byref(c), r * 6 + col, player, use_marks, C.byref(got)) == 0
                        assert got.value == want, (trial, use_marks, player, r, col)
    got = C.c_int32(7)
    assert fn["lz_scalar_piece_in_shape"](C.byref(c), 36, 1, 0, C.byref(got)) == 0 and got.value == 0       # off the board
    assert fn["lz_scalar_piece_in_shape"](C.byref(c), 0, 2, 0, C.byref(got)) == -1                         # LZ_ERR_ARG
