"""The scalar half of the reference's `v0_core` module: `GameState`, `MoveRecord`, `ActionCode`, `TensorStateBatch`, the
`Player` / `ActionType` enums and the one-state-at-a-time rule functions (v0/src/bindings/module.cpp:877-1156).

`v0/python/move_generator.py` (the C++-backed stand-in for `src.move_generator`), the rule tests and the tools of the
reference call these on the host, one state per call; they are host functions there (v0/src/rules/rule_engine.cpp,
v0/src/moves/move_generator.cpp) and here: every function goes through the C ABI of include/liuzhou_scalar.h in
libliuzhou_host.so (csrc/lz_scalar.cpp: the bitboard rules the gfx950 kernels include).  Same names, argument names,
defaults, return shapes ((row, col) tuples, ((r, c), (r, c)) movement pairs, lists in the reference's order) and the same
error behaviour: what the reference throws as std::runtime_error is a RuntimeError here.  Nothing on the self-play path
uses this module; without the host library it fails loudly (no Python re-implementation of the rules behind it)."""
from __future__ import annotations

import ctypes as C
import enum
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

Coord = Tuple[int, int]
Move = Tuple[Coord, Coord]
BOARD = 6
CELLS = 36


class Phase(enum.IntEnum):
    """module.cpp:877-885"""
    PLACEMENT = 1
    MARK_SELECTION = 2
    REMOVAL = 3
    MOVEMENT = 4
    CAPTURE_SELECTION = 5
    FORCED_REMOVAL = 6
    COUNTER_REMOVAL = 7


class Player(enum.IntEnum):
    """module.cpp:887-890"""
    BLACK = 1
    WHITE = -1


class ActionType(enum.IntEnum):
    """module.cpp:892-901"""
    PLACE = 1
    MOVE = 2
    MARK = 3
    CAPTURE = 4
    FORCED_REMOVAL = 5
    COUNTER_REMOVAL = 6
    NO_MOVES_REMOVAL = 7
    PROCESS_REMOVAL = 8


_ACTION_NAMES = {ActionType.PLACE: "place", ActionType.MOVE: "move", ActionType.MARK: "mark",
                 ActionType.CAPTURE: "capture", ActionType.FORCED_REMOVAL: "remove",
                 ActionType.COUNTER_REMOVAL: "counter_remove", ActionType.NO_MOVES_REMOVAL: "no_moves_remove",
                 ActionType.PROCESS_REMOVAL: "process_removal"}      # move_generator.cpp:50-71


# ---- the C structs of include/liuzhou_scalar.h --------------------------------------------------------------------
class _CState(C.Structure):
    _fields_ = [("board", C.c_int8 * CELLS), ("marks_black", C.c_uint8 * CELLS), ("marks_white", C.c_uint8 * CELLS),
                ("phase", C.c_int32), ("current_player", C.c_int32), ("forced_removals_done", C.c_int32),
                ("move_count", C.c_int32), ("pending_marks_required", C.c_int32),
                ("pending_marks_remaining", C.c_int32), ("pending_captures_required", C.c_int32),
                ("pending_captures_remaining", C.c_int32), ("moves_since_capture", C.c_int32)]


class _CMove(C.Structure):
    _fields_ = [("phase", C.c_int32), ("action_type", C.c_int32), ("primary", C.c_int32), ("secondary", C.c_int32)]


(_LIST_PLACEMENT, _LIST_MARKS, _LIST_MOVEMENT, _LIST_CAPTURES, _LIST_FORCED, _LIST_NO_MOVES, _LIST_COUNTER,
 _LIST_ALL) = range(8)
(_STEP_PLACEMENT, _STEP_MARK, _STEP_PROCESS, _STEP_MOVEMENT, _STEP_CAPTURE, _STEP_FORCED, _STEP_NO_MOVES,
 _STEP_COUNTER) = range(1, 9)
_ERR_ILLEGAL = -5

# LzScalarReason -> the wording of the error (the reference's messages are its own; only the type is contract)
_REASONS = {
    1: "the state is not in the phase this function serves",
    2: "position outside the board",
    3: "the cell is occupied",
    4: "the cell is marked by the opponent",
    5: "no pending mark / capture left",
    6: "the target is not a piece of the side that has to lose one",
    7: "the piece is already marked",
    8: "a piece of a square / line cannot be chosen here",
    9: "the start cell does not hold a piece of the player to move",
    10: "a move is one orthogonal step",
    11: "forced removal out of order",
    12: "Move phase does not match state phase.",
    13: "the action type is not allowed in this phase",
}

_SYMS: Dict[str, Any] = {}


def _host():
    """libliuzhou_host.so with the scalar entry points typed (raises if the library cannot be built / loaded)."""
    if not _SYMS:
        H = L.host_lib()
        i32p = C.POINTER(C.c_int32)
        sp, mp = C.POINTER(_CState), C.POINTER(_CMove)
        protos = {
            "lz_scalar_generate": (sp, C.c_int, i32p, C.c_int32, i32p),
            "lz_scalar_has_movement": (sp, i32p, i32p),
            "lz_scalar_apply": (sp, C.c_int, C.c_int32, C.c_int32, sp, i32p),
            "lz_scalar_apply_move": (sp, mp, sp, i32p),
            "lz_scalar_status": (sp, i32p, i32p),
            "lz_scalar_piece_in_shape": (sp, C.c_int32, C.c_int32, C.c_int32, i32p),
        }
        for name, args in protos.items():
            fn = getattr(H, name)                    # AttributeError = a host library older than this module
            fn.restype, fn.argtypes = C.c_int, args
            _SYMS[name] = fn
    return _SYMS


def _check(status: int, reason: C.c_int32, what: str) -> None:
    if status == 0:
        return
    if status == _ERR_ILLEGAL:
        raise RuntimeError(f"{what}: {_REASONS.get(int(reason.value), 'illegal for this state')}")
    raise RuntimeError(f"{what}: {L._STATUS.get(status, status)}")


def _coord(pos: Any, what: str = "position") -> Coord:
    try:
        r, c = pos
        return int(r), int(c)
    except (TypeError, ValueError):
        raise TypeError(f"{what} must be a (row, col) pair") from None


def _cell(pos: Any) -> int:
    r, c = _coord(pos)
    return r * BOARD + c if (0 <= r < BOARD and 0 <= c < BOARD) else -1


def _rc(cell: int) -> Coord:
    return (cell // BOARD, cell % BOARD)


# ---- GameState -------------------------------------------------------------------------------------------------
class GameState:
    """module.cpp:974-1006 over v0::GameState (v0/include/v0/game_state.hpp:97-147): `board` is a 6x6 nested list,
    `marked_black` / `marked_white` lists of (row, col) in ascending cell order; `moves_since_capture` is carried (and
    advanced by `apply_move_struct`) but, as in the reference, not a Python attribute of the public surface."""

    __slots__ = ("_c",)

    def __init__(self) -> None:
        self._c = _CState()
        self._c.phase = int(Phase.PLACEMENT)
        self._c.current_player = int(Player.BLACK)

    @classmethod
    def _wrap(cls, c: _CState) -> "GameState":
        g = cls.__new__(cls)
        g._c = c
        return g

    # board / marks
    @property
    def board(self) -> List[List[int]]:
        b = self._c.board
        return [[int(b[r * BOARD + c]) for c in range(BOARD)] for r in range(BOARD)]

    @board.setter
    def board(self, rows: Sequence[Sequence[int]]) -> None:
        rows = [list(r) for r in rows]
        if len(rows) != BOARD or any(len(r) != BOARD for r in rows):
            raise RuntimeError("board must be a 6x6 list")                       # module.cpp:43-49
        vals = [int(v) for r in rows for v in r]
        if any(v < -1 or v > 1 for v in vals):
            raise RuntimeError("board values must be in [-1, 1]")                # module.cpp:52-54
        for i, v in enumerate(vals):
            self._c.board[i] = v

    def _marks(self, arr) -> List[Coord]:
        return [_rc(i) for i in range(CELLS) if arr[i]]

    def _set_marks(self, arr, coords) -> None:
        cells = []
        for pos in list(coords):
            cell = _cell(pos)
            if cell < 0:
                raise RuntimeError("mark coordinates outside the board")        # module.cpp:70-73
            cells.append(cell)
        for i in range(CELLS):
            arr[i] = 0
        for cell in cells:
            arr[cell] = 1

    @property
    def marked_black(self) -> List[Coord]:
        return self._marks(self._c.marks_black)

    @marked_black.setter
    def marked_black(self, coords) -> None:
        self._set_marks(self._c.marks_black, coords)

    @property
    def marked_white(self) -> List[Coord]:
        return self._marks(self._c.marks_white)

    @marked_white.setter
    def marked_white(self, coords) -> None:
        self._set_marks(self._c.marks_white, coords)

    # enum fields
    @property
    def phase(self) -> Phase:
        return Phase(int(self._c.phase))

    @phase.setter
    def phase(self, v) -> None:
        self._c.phase = int(Phase(int(v)))

    @property
    def current_player(self) -> Player:
        return Player(int(self._c.current_player))

    @current_player.setter
    def current_player(self, v) -> None:
        self._c.current_player = int(Player(int(v)))

    def copy(self) -> "GameState":
        c = _CState()
        C.memmove(C.byref(c), C.byref(self._c), C.sizeof(_CState))
        return GameState._wrap(c)

    __copy__ = copy

    def __deepcopy__(self, memo) -> "GameState":
        return self.copy()

    def switch_player(self) -> None:
        self._c.current_player = -int(self._c.current_player)

    def is_board_full(self) -> bool:
        return all(self._c.board[i] != 0 for i in range(CELLS))

    def count_player_pieces(self, player) -> int:
        v = int(Player(int(player)))
        return sum(1 for i in range(CELLS) if self._c.board[i] == v)

    def get_player_pieces(self, player) -> List[Coord]:
        v = int(Player(int(player)))
        return [_rc(i) for i in range(CELLS) if self._c.board[i] == v]

    def __repr__(self) -> str:
        return (f"<v0_core.GameState phase={self.phase.name} player={self.current_player.name} "
                f"move_count={self.move_count}>")


def _int_field(name: str):
    def get(self) -> int:
        return int(getattr(self._c, name))

    def put(self, v) -> None:
        setattr(self._c, name, int(v))
    return property(get, put)


for _name in ("forced_removals_done", "move_count", "pending_marks_required", "pending_marks_remaining",
              "pending_captures_required", "pending_captures_remaining"):
    setattr(GameState, _name, _int_field(_name))


def _state_c(state: Any) -> _CState:
    """A GameState, or any object with its attributes (`src.game_state.GameState`: module.cpp:79-176 CoerceGameStateLike)."""
    if isinstance(state, GameState):
        return state._c
    if state is None:
        raise RuntimeError("state must not be None")
    g = GameState()
    try:
        g.board = [list(r) for r in state.board]
        ph, pl = state.phase, state.current_player
    except AttributeError as e:
        raise RuntimeError(f"state lacks attribute {e}") from None
    g.phase = int(getattr(ph, "value", ph))
    g.current_player = int(getattr(pl, "value", pl))
    for attr, setter in (("marked_black", "marked_black"), ("marked_white", "marked_white")):
        marks = getattr(state, attr, None)
        setattr(g, setter, [] if marks is None else sorted(tuple(m) for m in marks))
    for attr in ("forced_removals_done", "move_count", "pending_marks_required", "pending_marks_remaining",
                 "pending_captures_required", "pending_captures_remaining"):
        v = getattr(state, attr, None)
        if v is not None:
            setattr(g, attr, int(v))
    return g._c


# ---- MoveRecord / ActionCode ------------------------------------------------------------------------------------
class MoveRecord:
    """module.cpp:903-961 over v0::MoveRecord (move_generator.hpp:25-47); built by the static factories only."""

    __slots__ = ("_phase", "_type", "_primary", "_secondary")

    def __init__(self, *_a, **_k) -> None:
        raise TypeError("v0_core.MoveRecord: No constructor defined!")            # what PyBind11 says for a class without init

    @classmethod
    def _make(cls, phase: int, action_type: int, primary: Coord = (-1, -1), secondary: Coord = (-1, -1)) -> "MoveRecord":
        m = object.__new__(cls)
        m._phase, m._type, m._primary, m._secondary = Phase(int(phase)), ActionType(int(action_type)), primary, secondary
        return m

    @staticmethod
    def _valid(pos: Coord) -> bool:
        return 0 <= pos[0] < BOARD and 0 <= pos[1] < BOARD

    phase = property(lambda self: self._phase)
    action_type = property(lambda self: self._type)
    action_type_name = property(lambda self: _ACTION_NAMES[self._type])

    @property
    def position(self) -> Optional[Coord]:
        ok = self._type not in (ActionType.MOVE, ActionType.PROCESS_REMOVAL) and self._valid(self._primary)
        return self._primary if ok else None

    @property
    def from_position(self) -> Optional[Coord]:
        return self._primary if self._type == ActionType.MOVE and self._valid(self._primary) else None

    @property
    def to_position(self) -> Optional[Coord]:
        return self._secondary if self._type == ActionType.MOVE and self._valid(self._secondary) else None

    def to_dict(self) -> Dict[str, Any]:
        d: Dict[str, Any] = {"phase": self._phase, "action_type": _ACTION_NAMES[self._type]}
        if self._type == ActionType.MOVE:
            d["from_position"], d["to_position"] = self._primary, self._secondary
        elif self.position is not None:
            d["position"] = self._primary
        return d

    @staticmethod
    def placement(position) -> "MoveRecord":
        return MoveRecord._make(Phase.PLACEMENT, ActionType.PLACE, _coord(position))

    @staticmethod
    def mark(position) -> "MoveRecord":
        return MoveRecord._make(Phase.MARK_SELECTION, ActionType.MARK, _coord(position))

    @staticmethod
    def capture(position) -> "MoveRecord":
        return MoveRecord._make(Phase.CAPTURE_SELECTION, ActionType.CAPTURE, _coord(position))

    @staticmethod
    def forced_removal(position) -> "MoveRecord":
        return MoveRecord._make(Phase.FORCED_REMOVAL, ActionType.FORCED_REMOVAL, _coord(position))

    @staticmethod
    def counter_removal(position) -> "MoveRecord":
        return MoveRecord._make(Phase.COUNTER_REMOVAL, ActionType.COUNTER_REMOVAL, _coord(position))

    @staticmethod
    def no_moves_removal(position) -> "MoveRecord":
        return MoveRecord._make(Phase.MOVEMENT, ActionType.NO_MOVES_REMOVAL, _coord(position))

    @staticmethod
    def process_removal() -> "MoveRecord":
        return MoveRecord._make(Phase.REMOVAL, ActionType.PROCESS_REMOVAL)

    @staticmethod
    def movement(from_position, to_position) -> "MoveRecord":
        return MoveRecord._make(Phase.MOVEMENT, ActionType.MOVE, _coord(from_position, "from_position"),
                                _coord(to_position, "to_position"))

    def __repr__(self) -> str:
        return f"<v0_core.MoveRecord {self.to_dict()}>"


class ActionCode:
    """module.cpp:962-972 over v0::ActionCode (move_generator.hpp:49-54)."""

    __slots__ = ("kind", "primary", "secondary", "extra")

    def __init__(self) -> None:
        self.kind = self.primary = self.secondary = self.extra = 0

    def to_tuple(self) -> Tuple[int, int, int, int]:
        return (int(self.kind), int(self.primary), int(self.secondary), int(self.extra))

    def __repr__(self) -> str:
        return f"<v0_core.ActionCode {self.to_tuple()}>"


# ---- rule functions ----------------------------------------------------------------------------------------------
def _listing(state: Any, what: int) -> List[int]:
    buf = (C.c_int32 * 640)()
    n = C.c_int32(0)
    st = _host()["lz_scalar_generate"](C.byref(_state_c(state)), what, buf, 640, C.byref(n))
    _check(st, C.c_int32(0), "generate")
    return list(buf[: int(n.value)])


def _cells(state: Any, what: int) -> List[Coord]:
    return [_rc(c) for c in _listing(state, what)]


def _records(state: Any, what: int) -> List[MoveRecord]:
    v = _listing(state, what)
    return [MoveRecord._make(v[i], v[i + 1], _rc(v[i + 2]) if v[i + 2] >= 0 else (-1, -1),
                             _rc(v[i + 3]) if v[i + 3] >= 0 else (-1, -1)) for i in range(0, len(v), 4)]


def _step(state: Any, what: int, a: int = -1, b: int = -1, name: str = "apply") -> GameState:
    out, why = _CState(), C.c_int32(0)
    st = _host()["lz_scalar_apply"](C.byref(_state_c(state)), what, a, b, C.byref(out), C.byref(why))
    _check(st, why, name)
    return GameState._wrap(out)


def generate_placement_positions(state) -> List[Coord]:
    """module.cpp:1008 -> rule_engine.cpp:210-224"""
    return _cells(state, _LIST_PLACEMENT)


def apply_placement_move(state, position) -> GameState:
    """module.cpp:1009 -> rule_engine.cpp:226-279"""
    return _step(state, _STEP_PLACEMENT, _cell(position), name="apply_placement_move")


def generate_mark_targets(state) -> List[Coord]:
    """module.cpp:1011 -> rule_engine.cpp:281-308"""
    return _cells(state, _LIST_MARKS)


def apply_mark_selection(state, position) -> GameState:
    """module.cpp:1012 -> rule_engine.cpp:310-358"""
    return _step(state, _STEP_MARK, _cell(position), name="apply_mark_selection")


def process_phase2_removals(state) -> GameState:
    """module.cpp:1014 -> rule_engine.cpp:360-395"""
    return _step(state, _STEP_PROCESS, name="process_phase2_removals")


def generate_movement_moves(state) -> List[Move]:
    """module.cpp:1016 -> rule_engine.cpp:397-419"""
    v = _listing(state, _LIST_MOVEMENT)
    return [(_rc(v[i]), _rc(v[i + 1])) for i in range(0, len(v), 2)]


def has_legal_movement_moves(state) -> bool:
    """module.cpp:1017 -> rule_engine.cpp:421-427 (raises outside the movement phase)"""
    has, why = C.c_int32(0), C.c_int32(0)
    st = _host()["lz_scalar_has_movement"](C.byref(_state_c(state)), C.byref(has), C.byref(why))
    _check(st, why, "has_legal_movement_moves")
    return bool(has.value)


def _move_pair(move) -> Tuple[int, int]:
    try:
        a, b = move
    except (TypeError, ValueError):
        raise TypeError("move must be a ((row, col), (row, col)) pair") from None
    return _cell(a), _cell(b)


def apply_movement_move(state, move, quiet: bool = False) -> GameState:
    """module.cpp:1018-1023 -> rule_engine.cpp:429-479"""
    a, b = _move_pair(move)
    return _step(state, _STEP_MOVEMENT, a, b, name="apply_movement_move")


def generate_capture_targets(state) -> List[Coord]:
    """module.cpp:1025 -> rule_engine.cpp:481-499"""
    return _cells(state, _LIST_CAPTURES)


def apply_capture_selection(state, position, quiet: bool = False) -> GameState:
    """module.cpp:1026-1031 -> rule_engine.cpp:501-547"""
    return _step(state, _STEP_CAPTURE, _cell(position), name="apply_capture_selection")


def apply_forced_removal(state, piece_to_remove) -> GameState:
    """module.cpp:1033-1037 -> rule_engine.cpp:549-595"""
    return _step(state, _STEP_FORCED, _cell(piece_to_remove), name="apply_forced_removal")


def handle_no_moves_phase3(state, stucked_player_removes, quiet: bool = False) -> GameState:
    """module.cpp:1038-1043 -> rule_engine.cpp:597-637"""
    return _step(state, _STEP_NO_MOVES, _cell(stucked_player_removes), name="handle_no_moves_phase3")


def apply_counter_removal_phase3(state, opponent_removes, quiet: bool = False) -> GameState:
    """module.cpp:1044-1049 -> rule_engine.cpp:639-680"""
    return _step(state, _STEP_COUNTER, _cell(opponent_removes), name="apply_counter_removal_phase3")


def generate_legal_moves_phase1(state) -> List[Coord]:
    """module.cpp:1051 -> rule_engine.cpp:682-684"""
    return generate_placement_positions(state)


def apply_move_phase1(state, move, mark_positions=None) -> GameState:
    """module.cpp:1052-1064 -> rule_engine.cpp:686-700: the placement, then the marks it earns"""
    nxt = apply_placement_move(state, move)
    marks = [] if mark_positions is None else list(mark_positions)
    if marks:
        if nxt.phase != Phase.MARK_SELECTION:
            raise RuntimeError("apply_move_phase1: mark_positions given but the placement earns no mark")
        for pos in marks:
            nxt = apply_mark_selection(nxt, pos)
    return nxt


def generate_legal_moves_phase3(state) -> List[Move]:
    """module.cpp:1066 -> rule_engine.cpp:702-704"""
    return generate_movement_moves(state)


def has_legal_moves_phase3(state) -> bool:
    """module.cpp:1067 -> rule_engine.cpp:706-708"""
    return has_legal_movement_moves(state)


def apply_move_phase3(state, move, capture_positions=None, quiet: bool = False) -> GameState:
    """module.cpp:1068-1082 -> rule_engine.cpp:710-725: the step, then the captures it earns"""
    nxt = apply_movement_move(state, move, quiet)
    caps = [] if capture_positions is None else list(capture_positions)
    if caps:
        if nxt.phase != Phase.CAPTURE_SELECTION:
            raise RuntimeError("apply_move_phase3: capture_positions given but the move earns no capture")
        for pos in caps:
            nxt = apply_capture_selection(nxt, pos, quiet)
    return nxt


def generate_all_legal_moves_struct(state) -> List[MoveRecord]:
    """module.cpp:1083-1086 -> move_generator.cpp:242-297 (empty once the game is over)"""
    return _records(state, _LIST_ALL)


def generate_forced_removal_moves_struct(state) -> List[MoveRecord]:
    """module.cpp:1091-1094 -> move_generator.cpp:149-175"""
    return _records(state, _LIST_FORCED)


def generate_no_moves_options_struct(state) -> List[MoveRecord]:
    """module.cpp:1095-1098 -> move_generator.cpp:177-207"""
    return _records(state, _LIST_NO_MOVES)


def generate_counter_removal_moves_struct(state) -> List[MoveRecord]:
    """module.cpp:1099-1102 -> move_generator.cpp:209-240"""
    return _records(state, _LIST_COUNTER)


def encode_action_code(move: MoveRecord) -> ActionCode:
    """module.cpp:1104 -> move_generator.cpp:299-349: (kind, primary cell, secondary cell, 0); kind == action type"""
    if not isinstance(move, MoveRecord):
        raise TypeError("encode_action_code: expected a MoveRecord")
    code = ActionCode()
    code.kind = int(move.action_type)
    cell = lambda pos: pos[0] * BOARD + pos[1]
    if move.action_type == ActionType.MOVE:
        code.primary, code.secondary = cell(move._primary), cell(move._secondary)
    elif move.action_type != ActionType.PROCESS_REMOVAL:
        code.primary = cell(move._primary)
    return code


def encode_action_codes(moves: Sequence[MoveRecord]) -> List[ActionCode]:
    """module.cpp:1103 -> move_generator.cpp:351-358"""
    return [encode_action_code(m) for m in moves]


def generate_moves_with_codes(state) -> Tuple[List[MoveRecord], List[ActionCode]]:
    """module.cpp:1087-1090 -> move_generator.cpp:434-439"""
    moves = generate_all_legal_moves_struct(state)
    return moves, encode_action_codes(moves)


def apply_move_struct(state, move: MoveRecord, quiet: bool = False) -> GameState:
    """module.cpp:1105-1110 -> move_generator.cpp:360-432: the transition the record names, `move_count + 1`, and the
    no-capture counter (0 after a placement / mark or when a piece left the board, else + 1)"""
    if not isinstance(move, MoveRecord):
        raise TypeError("apply_move_struct: expected a MoveRecord")
    cell = lambda pos: pos[0] * BOARD + pos[1] if MoveRecord._valid(pos) else -1
    m = _CMove(int(move.phase), int(move.action_type), cell(move._primary), cell(move._secondary))
    out, why = _CState(), C.c_int32(0)
    st = _host()["lz_scalar_apply_move"](C.byref(_state_c(state)), C.byref(m), C.byref(out), C.byref(why))
    _check(st, why, "apply_move_struct")
    return GameState._wrap(out)


# ---- TensorStateBatch ----------------------------------------------------------------------------------------------
_BATCH_FIELDS = ("board", "marks_black", "marks_white", "phase", "current_player", "pending_marks_required",
                 "pending_marks_remaining", "pending_captures_required", "pending_captures_remaining",
                 "forced_removals_done", "move_count", "moves_since_capture", "mask_alive")


class TensorStateBatch:
    """module.cpp:1112-1144 over v0::TensorStateBatch (v0/include/v0/tensor_state_batch.hpp:11-35): 13 tensors,
    read-only attributes; `moves_since_capture` travels inside (the reference binds no property for it)."""

    def __init__(self) -> None:
        self._t: Dict[str, Optional[torch.Tensor]] = {k: None for k in _BATCH_FIELDS}
        self._board_size = BOARD

    @classmethod
    def _of(cls, tensors: Dict[str, torch.Tensor]) -> "TensorStateBatch":
        b = cls()
        b._t = dict(tensors)
        return b

    board_size = property(lambda self: self._board_size)

    def device(self) -> torch.device:
        return self._t["board"].device

    def to(self, device: str) -> "TensorStateBatch":
        dev = _device(device)
        return TensorStateBatch._of({k: v.to(dev) for k, v in self._t.items()})

    def clone(self) -> "TensorStateBatch":
        return TensorStateBatch._of({k: v.clone() for k, v in self._t.items()})


for _name in _BATCH_FIELDS:
    if _name != "moves_since_capture":
        setattr(TensorStateBatch, _name, property(lambda self, _k=_name: self._t[_k]))


def _device(device) -> torch.device:
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", 0)                                            # tensor_state_batch.cpp:22-27
    return dev


def tensor_batch_from_game_states(states: Sequence[GameState], device: str = "cpu") -> TensorStateBatch:
    """module.cpp:1146-1152 -> tensor_state_batch.cpp:69-172: staged in (pinned, for a device target) host tensors"""
    states = list(states)
    if not states:
        raise RuntimeError("from_game_states requires at least one GameState instance.")
    dev = _device(device)
    cs = [_state_c(s) for s in states]
    n = len(cs)
    pin = dev.type == "cuda"
    mk = lambda shape, dt, fill=0: torch.full(shape, fill, dtype=dt, pin_memory=pin)
    board = mk((n, BOARD, BOARD), torch.int8)
    mb, mw = mk((n, BOARD, BOARD), torch.bool), mk((n, BOARD, BOARD), torch.bool)
    scal = {k: mk((n,), torch.int64) for k in _BATCH_FIELDS[3:12]}
    bb, bmb, bmw = board.view(n, CELLS).numpy(), mb.view(n, CELLS).numpy(), mw.view(n, CELLS).numpy()
    cols = {k: v.numpy() for k, v in scal.items()}
    for i, c in enumerate(cs):
        bb[i, :] = c.board[:]
        bmb[i, :] = [bool(x) for x in c.marks_black]
        bmw[i, :] = [bool(x) for x in c.marks_white]
        cols["phase"][i] = c.phase
        cols["current_player"][i] = 1 if c.current_player >= 0 else -1
        for k in _BATCH_FIELDS[5:12]:
            cols[k][i] = getattr(c, k)
    host = {"board": board, "marks_black": mb, "marks_white": mw, **scal,
            "mask_alive": mk((n,), torch.bool, True)}
    if dev.type != "cuda":
        return TensorStateBatch._of(host)
    return TensorStateBatch._of({k: v.to(dev, non_blocking=True) for k, v in host.items()})


def tensor_batch_to_game_states(batch: TensorStateBatch) -> List[GameState]:
    """module.cpp:1153-1156 -> tensor_state_batch.cpp:174-262"""
    if not isinstance(batch, TensorStateBatch):
        raise TypeError("tensor_batch_to_game_states: expected a TensorStateBatch")
    t = batch._t
    n = int(t["board"].shape[0])
    cpu = lambda x, dt: x.to("cpu", dt).contiguous()
    board = cpu(t["board"], torch.int8).view(n, CELLS).numpy()
    mb = cpu(t["marks_black"], torch.bool).view(n, CELLS).numpy()
    mw = cpu(t["marks_white"], torch.bool).view(n, CELLS).numpy()
    cols = {k: cpu(t[k], torch.int64).numpy() for k in _BATCH_FIELDS[3:12]}
    out = []
    for i in range(n):
        c = _CState()
        for j in range(CELLS):
            c.board[j] = int(board[i, j])
            c.marks_black[j] = 1 if mb[i, j] else 0
            c.marks_white[j] = 1 if mw[i, j] else 0
        c.phase = int(cols["phase"][i])
        c.current_player = 1 if int(cols["current_player"][i]) >= 0 else -1
        for k in _BATCH_FIELDS[5:12]:
            setattr(c, k, int(cols[k][i]))
        out.append(GameState._wrap(c))
    return out


__all__ = ["Phase", "Player", "ActionType", "MoveRecord", "ActionCode", "GameState", "TensorStateBatch",
           "generate_placement_positions", "apply_placement_move", "generate_mark_targets", "apply_mark_selection",
           "process_phase2_removals", "generate_movement_moves", "has_legal_movement_moves", "apply_movement_move",
           "generate_capture_targets", "apply_capture_selection", "apply_forced_removal", "handle_no_moves_phase3",
           "apply_counter_removal_phase3", "generate_legal_moves_phase1", "apply_move_phase1",
           "generate_legal_moves_phase3", "has_legal_moves_phase3", "apply_move_phase3",
           "generate_all_legal_moves_struct", "generate_moves_with_codes", "generate_forced_removal_moves_struct",
           "generate_no_moves_options_struct", "generate_counter_removal_moves_struct", "encode_action_codes",
           "encode_action_code", "apply_move_struct", "tensor_batch_from_game_states", "tensor_batch_to_game_states"]

# `export_values()` of the three enums (module.cpp:885,890,901): the members are module attributes too; where two enums
# share a name (FORCED_REMOVAL, COUNTER_REMOVAL) the one bound last -- ActionType -- is what the attribute holds
EXPORTED_VALUES: Dict[str, Any] = {}
for _enum in (Phase, Player, ActionType):
    for _member in _enum:
        EXPORTED_VALUES[_member.name] = _member
globals().update(EXPORTED_VALUES)
__all__ += list(EXPORTED_VALUES)
