"""A deterministic walk through the scalar `v0_core` surface (GameState, the generate_* / apply_* functions, MoveRecord,
ActionCode, TensorStateBatch), recorded as flat integer arrays.

`record(core, seed)` drives any module with that surface -- the reference's own compiled `v0_core` (oracle/_ref, by
`oracle/gen_golden_scalar.py`, which commits the result as tests/golden/g18_scalar_surface.npz) or ours
(tests/test_scalar_surface.py) -- through: seeded random games played to the end with `apply_move_struct` (so the hidden
no-capture counter is exercised through the draw rule), crafted starts (forced removal, a stuck player), and random
unreachable states; at every state it records every list the module generates and the outcome of a fixed set of probe
calls (most of them illegal: raised or not, and the state that came back).  Two modules with the same behaviour give
byte-identical records."""
import numpy as np

FIELDS = ("forced_removals_done", "move_count", "pending_marks_required", "pending_marks_remaining",
          "pending_captures_required", "pending_captures_remaining")
LISTS = ("generate_placement_positions", "generate_mark_targets", "generate_movement_moves", "generate_capture_targets",
         "generate_forced_removal_moves_struct", "generate_no_moves_options_struct",
         "generate_counter_removal_moves_struct", "generate_all_legal_moves_struct", "generate_legal_moves_phase1",
         "generate_legal_moves_phase3")
STEPS = ("apply_placement_move", "apply_mark_selection", "apply_capture_selection", "apply_forced_removal",
         "handle_no_moves_phase3", "apply_counter_removal_phase3")


def state_row(s):
    """The public face of a GameState as 117 small integers."""
    board = [v for row in s.board for v in row]
    mb, mw = [0] * 36, [0] * 36
    for r, c in s.marked_black:
        mb[r * 6 + c] = 1
    for r, c in s.marked_white:
        mw[r * 6 + c] = 1
    return board + mb + mw + [int(s.phase.value), int(s.current_player.value)] + [int(getattr(s, f)) for f in FIELDS] + [0]


def move_row(m):
    p, f, t = m.position, m.from_position, m.to_position
    enc = lambda x: -1 if x is None else x[0] * 6 + x[1]
    return [int(m.phase.value), int(m.action_type.value), enc(p), enc(f), enc(t)]


def flat_list(name, items):
    out = []
    for it in items:
        if name.endswith("_struct"):
            out += move_row(it)
        elif isinstance(it[0], tuple):
            out += [it[0][0] * 6 + it[0][1], it[1][0] * 6 + it[1][1]]
        else:
            out += [it[0] * 6 + it[1]]
    return out


class Recorder:
    def __init__(self, core):
        self.core = core
        self.states, self.lists, self.list_off, self.probes, self.codes, self.tags = [], [], [0], [], [], []

    def outcome(self, fn, *args, **kw):
        """[raised, state row...] of a call that returns a GameState, or [raised, value]."""
        try:
            r = fn(*args, **kw)
        except RuntimeError:
            return [1] + [0] * 117
        if isinstance(r, bool):
            return [0, int(r)] + [0] * 116
        return [0] + state_row(r)

    def visit(self, s, rng, tag):
        core = self.core
        self.tags.append(tag)
        self.states.append(state_row(s))
        for name in LISTS:
            self.lists += flat_list(name, getattr(core, name)(s))
            self.list_off.append(len(self.lists))
        moves, codes = core.generate_moves_with_codes(s)
        assert len(moves) == len(codes)
        for m, c in zip(moves, codes):
            self.codes.append(list(c.to_tuple()) + list(core.encode_action_code(m).to_tuple()))
            d = m.to_dict()
            assert d["action_type"] == m.action_type_name and d["phase"] == m.phase
        # probe calls: every single-cell transition with three random cells (one of them off the board now and then),
        # the movement step with a random pair and with a neighbouring pair, the composite phase-1 / phase-3 calls
        cells = [(int(rng.integers(-1, 7)), int(rng.integers(0, 6))) for _ in range(2)] + [(int(rng.integers(0, 6)), int(rng.integers(0, 6)))]
        for name in STEPS:
            for pos in cells:
                self.probes.append(self.outcome(getattr(core, name), s, pos))
        a = (int(rng.integers(0, 6)), int(rng.integers(0, 6)))
        d = [(-1, 0), (1, 0), (0, -1), (0, 1)][int(rng.integers(0, 4))]
        for b in ((a[0] + d[0], a[1] + d[1]), (int(rng.integers(0, 6)), int(rng.integers(0, 6)))):
            self.probes.append(self.outcome(core.apply_movement_move, s, (a, b)))
        self.probes.append(self.outcome(core.process_phase2_removals, s))
        self.probes.append(self.outcome(core.has_legal_movement_moves, s))
        self.probes.append(self.outcome(core.has_legal_moves_phase3, s))
        self.probes.append(self.outcome(core.apply_move_phase1, s, cells[2], None))
        self.probes.append(self.outcome(core.apply_move_phase1, s, cells[2], [cells[0]]))
        self.probes.append(self.outcome(core.apply_move_phase3, s, (a, (a[0] + d[0], a[1] + d[1])), [cells[2]]))
        # the phase's own function with legal arguments (up to three of the legal moves)
        direct = {"place": core.apply_placement_move, "mark": core.apply_mark_selection, "capture": core.apply_capture_selection,
                  "remove": core.apply_forced_removal, "no_moves_remove": core.handle_no_moves_phase3,
                  "counter_remove": core.apply_counter_removal_phase3}
        for k in sorted(set(int(x) for x in rng.integers(0, max(1, len(moves)), size=3)))[:len(moves)]:
            m = moves[k]
            if m.action_type_name == "move":
                self.probes.append(self.outcome(core.apply_movement_move, s, (m.from_position, m.to_position)))
            elif m.action_type_name == "process_removal":
                self.probes.append(self.outcome(core.process_phase2_removals, s))
            else:
                self.probes.append(self.outcome(direct[m.action_type_name], s, m.position))
        # a record of the wrong phase / type
        wrong = core.MoveRecord.capture(cells[2]) if s.phase != core.Phase.CAPTURE_SELECTION else core.MoveRecord.placement(cells[2])
        self.probes.append(self.outcome(core.apply_move_struct, s, wrong))
        return moves

    def play(self, s, rng, tag, max_steps=400):
        core = self.core
        for _ in range(max_steps):
            moves = self.visit(s, rng, tag)
            if not moves:
                return
            s = core.apply_move_struct(s, moves[int(rng.integers(0, len(moves)))], quiet=True)

    def arrays(self):
        return {"states": np.asarray(self.states, dtype=np.int16), "lists": np.asarray(self.lists, dtype=np.int16),
                "list_off": np.asarray(self.list_off, dtype=np.int64), "probes": np.asarray(self.probes, dtype=np.int16),
                "codes": np.asarray(self.codes, dtype=np.int16).reshape(-1, 8), "tags": np.asarray(self.tags, dtype=np.int8)}


def crafted_states(core, rng):
    """Starts random play rarely reaches: a full board without marks (REMOVAL -> forced removals), players without a move."""
    out = []
    g = core.GameState()
    g.board = [[1 if ((r // 1 + c) % 2 == 0) else -1 for c in range(6)] for r in range(6)]      # chequered: no square, no line
    g.phase = core.Phase.REMOVAL
    g.move_count = 36
    out.append(g)
    g = core.GameState()
    rows = [[1, 1, -1, -1, 1, 1], [1, -1, -1, 1, 1, -1], [-1, 1, 1, -1, -1, 1], [1, 1, -1, -1, 1, 1], [-1, -1, 1, 1, -1, -1],
            [1, -1, -1, 1, 1, -1]]
    g.board = rows
    g.phase = core.Phase.REMOVAL
    g.marked_white = [(0, 2)]
    g.marked_black = [(5, 0), (3, 1)]
    g.move_count = 40
    out.append(g)
    tries = 0
    while len(out) < 10 and tries < 20000:                       # movement-phase boards on which the mover is stuck
        tries += 1
        cells = rng.integers(-1, 2, size=36)
        empties = rng.choice(36, size=int(rng.integers(1, 4)), replace=False)
        cells[cells == 0] = 1
        cells[empties] = 0
        g = core.GameState()
        g.board = cells.reshape(6, 6).tolist()
        g.phase = core.Phase.MOVEMENT
        g.current_player = core.Player.BLACK if rng.integers(0, 2) else core.Player.WHITE
        g.move_count = 50
        if not core.has_legal_movement_moves(g) and g.count_player_pieces(core.Player.BLACK) >= 4 and g.count_player_pieces(core.Player.WHITE) >= 4:
            out.append(g)
    return out


def garbage_state(core, rng):
    g = core.GameState()
    g.board = rng.choice([-1, 0, 1], size=(6, 6), p=[0.4, 0.2, 0.4]).tolist()
    pick = lambda k: [(int(c) // 6, int(c) % 6) for c in rng.choice(36, size=int(rng.integers(0, k)), replace=False)]
    g.marked_black, g.marked_white = pick(5), pick(5)
    g.phase = core.Phase(int(rng.integers(1, 8)))
    g.current_player = core.Player.BLACK if rng.integers(0, 2) else core.Player.WHITE
    g.forced_removals_done = int(rng.integers(0, 4))
    g.move_count = int(rng.integers(0, 150))
    g.pending_marks_required = int(rng.integers(0, 3)); g.pending_marks_remaining = int(rng.integers(0, 3))
    g.pending_captures_required = int(rng.integers(0, 3)); g.pending_captures_remaining = int(rng.integers(0, 3))
    return g


def record(core, seed=20261004, games=5, garbage=250):
    rng = np.random.default_rng(seed)
    rec = Recorder(core)
    for _ in range(games):
        rec.play(core.GameState(), rng, 0)
    for g in crafted_states(core, rng):
        rec.play(g, rng, 1, max_steps=40)
    for _ in range(garbage):
        g = garbage_state(core, rng)
        moves = rec.visit(g, rng, 2)
        for m in moves[:3]:                                       # a step from an unreachable state, through the record path
            rec.probes.append(rec.outcome(core.apply_move_struct, g, m, quiet=True))
    # the tensor batch of everything visited last: from / to GameState round trip
    last = [core.GameState()] + crafted_states(core, np.random.default_rng(seed + 1))[:4] + [garbage_state(core, rng) for _ in range(20)]
    batch = core.tensor_batch_from_game_states(last, "cpu")
    back = core.tensor_batch_to_game_states(batch.clone().to("cpu"))
    out = rec.arrays()
    out["batch_states"] = np.asarray([state_row(s) for s in last], dtype=np.int16)
    out["batch_back"] = np.asarray([state_row(s) for s in back], dtype=np.int16)
    for k in ("board", "marks_black", "marks_white", "phase", "current_player", "pending_marks_required",
              "pending_marks_remaining", "pending_captures_required", "pending_captures_remaining", "forced_removals_done",
              "move_count", "mask_alive"):
        out["batch_" + k] = getattr(batch, k).numpy().astype(np.int16)
    out["batch_board_size"] = np.asarray([batch.board_size])
    return out
