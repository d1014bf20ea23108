"""Self-play statistics record.  The field names and the `to_dict` payload are the interface the reference's
stage scripts and manifest readers consume (v1/python/self_play_types.py:9-60); here one schema table drives
both the dataclass and its serialiser, so a field cannot be added to one and forgotten in the other."""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Any, Callable, Dict, Tuple

_MISSING = object()


def _floats(m) -> Dict[str, float]:
    return {k: float(v) for k, v in m.items()}


def _ints(m) -> Dict[str, int]:
    return {k: int(v) for k, v in m.items()}


# (name, annotation, serialiser, default factory or _MISSING)
_SCHEMA: Tuple[Tuple[str, Any, Callable, Any], ...] = (
    *((n, int, float, _MISSING) for n in ("num_games", "num_positions", "black_wins", "white_wins", "draws")),
    *((n, float, float, _MISSING) for n in ("avg_game_length", "elapsed_sec", "positions_per_sec", "games_per_sec")),
    ("step_timing_ms", Dict[str, float], _floats, _MISSING),
    ("step_timing_ratio", Dict[str, float], _floats, _MISSING),
    ("step_timing_calls", Dict[str, int], _ints, _MISSING),
    ("mcts_counters", Dict[str, int], _ints, _MISSING),
    ("piece_delta_buckets", Dict[str, int], _ints, _MISSING),
    ("policy_target_audit", Dict[str, Any], lambda m: dict(m or {}), dict),
    ("device", str, str, str),
    ("fallback_count", int, int, int),
    ("fallback_reasons", Tuple[str, ...], list, tuple),
)


def _to_dict(self) -> Dict[str, object]:
    return {name: conv(getattr(self, name)) for name, _, conv, _ in _SCHEMA}


SelfPlayV1Stats = make_dataclass(
    "SelfPlayV1Stats",
    [(name, ann) if dflt is _MISSING else (name, ann, field(default_factory=dflt)) for name, ann, _, dflt in _SCHEMA],
    namespace={"to_dict": _to_dict, "__doc__": "Aggregate statistics of one self-play call."},
)
SelfPlayV1Stats.__module__ = __name__
