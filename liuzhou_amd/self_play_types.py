"""Self-play statistics record (same fields and `to_dict` payload as v1/python/self_play_types.py:9-60)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Tuple


@dataclass
class SelfPlayV1Stats:
    num_games: int
    num_positions: int
    black_wins: int
    white_wins: int
    draws: int
    avg_game_length: float
    elapsed_sec: float
    positions_per_sec: float
    games_per_sec: float
    step_timing_ms: Dict[str, float]
    step_timing_ratio: Dict[str, float]
    step_timing_calls: Dict[str, int]
    mcts_counters: Dict[str, int]
    piece_delta_buckets: Dict[str, int]
    policy_target_audit: Dict[str, Any] = field(default_factory=dict)
    device: str = ""
    fallback_count: int = 0
    fallback_reasons: Tuple[str, ...] = ()

    def to_dict(self) -> Dict[str, object]:
        d: Dict[str, object] = {k: float(getattr(self, k)) for k in (
            "num_games", "num_positions", "black_wins", "white_wins", "draws", "avg_game_length", "elapsed_sec",
            "positions_per_sec", "games_per_sec")}
        d["step_timing_ms"] = {k: float(v) for k, v in self.step_timing_ms.items()}
        d["step_timing_ratio"] = {k: float(v) for k, v in self.step_timing_ratio.items()}
        d["step_timing_calls"] = {k: int(v) for k, v in self.step_timing_calls.items()}
        d["mcts_counters"] = {k: int(v) for k, v in self.mcts_counters.items()}
        d["piece_delta_buckets"] = {k: int(v) for k, v in self.piece_delta_buckets.items()}
        d["policy_target_audit"] = dict(self.policy_target_audit or {})
        d["device"] = str(self.device)
        d["fallback_count"] = int(self.fallback_count)
        d["fallback_reasons"] = list(self.fallback_reasons)
        return d
