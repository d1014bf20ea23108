"""Evaluation arena on the self-play engine.

The reference arena (`scripts/eval_checkpoint.py:262-655`) keeps one Python `GameState` per game and rebuilds a GPU
batch for every move; here the games live on the device for their whole life: one `GpuStateBatch`, legal masks /
transitions / termination through the HIP operators, the two agents searching only the games in which they are to
move.  Semantics kept from the reference worker (`_eval_worker_v1`, :448-654):
  * the challenger plays black in the first half of the games and white in the second,
  * a side that has no legal move loses, the move limits give a draw, `opening_random_moves` plies (by `move_count`)
    are played uniformly at random by whoever is to move,
  * the opponent is another checkpoint (same search settings) or `RandomAgent` (uniform over the legal moves),
  * result payload = wins / losses / draws / rates from the challenger's side (+ per-colour breakdown).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch

from . import v0_core
from .mcts_gpu import GpuStateBatch, V1RootMCTS, V1RootMCTSConfig, encode_actions_fast
from .net_hip import FusedNet


@dataclass
class EvaluationStats:
    wins: int
    losses: int
    draws: int
    total_games: int
    color_breakdown: Dict[str, Dict[str, int]] = field(default_factory=dict)
    move_log: Optional[torch.Tensor] = None      # int32[games, plies] 220-d action indices (-1 pad), if recorded

    def _rate(self, v: int) -> float:
        return 0.0 if self.total_games == 0 else v / self.total_games

    @property
    def win_rate(self) -> float:
        return self._rate(self.wins)

    @property
    def loss_rate(self) -> float:
        return self._rate(self.losses)

    @property
    def draw_rate(self) -> float:
        return self._rate(self.draws)

    def to_payload(self, name: str) -> Dict[str, Any]:
        return {"name": str(name), "wins": int(self.wins), "losses": int(self.losses), "draws": int(self.draws),
                "total_games": int(self.total_games), "win_rate": float(self.win_rate),
                "loss_rate": float(self.loss_rate), "draw_rate": float(self.draw_rate),
                "color_breakdown": {k: dict(v) for k, v in self.color_breakdown.items()}}


def _uniform_legal_codes(state: GpuStateBatch):
    """Uniform random legal action per state -> (codes int32[B,4], valid bool[B], no_legal bool[B])."""
    mask, meta = encode_actions_fast(state)
    n = mask.sum(dim=1)
    valid = n > 0
    probs = mask.to(torch.float32)
    probs[~valid, 0] = 1.0                                  # keep multinomial well-defined; the row is flagged invalid
    idx = torch.multinomial(probs, 1).view(-1)
    codes = meta[torch.arange(mask.shape[0], device=mask.device), idx].to(torch.int32)
    codes[~valid] = -1
    return codes, valid, ~valid


def codes_to_indices(codes: torch.Tensor) -> torch.Tensor:
    """Action codes int32[N,4] (kind, primary, secondary, extra) -> 220-d action indices (v0/python/move_encoder.py:46-51:
    placement = cell, movement = 36 + 4*from + dir, selections = 180 + cell, process-removal = 216; invalid: -1)."""
    kind, a, b = codes[:, 0].to(torch.int64), codes[:, 1].to(torch.int64), codes[:, 2].to(torch.int64)
    idx = torch.full_like(kind, -1)
    idx = torch.where(kind == 1, a, idx)
    idx = torch.where(kind == 2, 36 + 4 * a + b, idx)
    idx = torch.where((kind >= 3) & (kind <= 7), 180 + a, idx)
    idx = torch.where(kind == 8, torch.full_like(kind, 216), idx)
    return idx.to(torch.int32)


class RandomAgent:
    """Uniform over the legal moves (the reference's vs-random opponent)."""

    def select(self, state: GpuStateBatch, force_uniform: Optional[torch.Tensor] = None):
        return _uniform_legal_codes(state)


class RootSearchAgent:
    """Checkpoint + the reference's evaluation search (V1RootMCTS, no root noise; eval_checkpoint.py:262-322)."""

    def __init__(self, model, device, mcts_simulations: int, temperature: float = 0.1, sample_moves: bool = False) -> None:
        dev = torch.device(device)
        from .net_hip import fused_supported
        net = FusedNet(model.to(dev).eval(), dev) if fused_supported(model) else model.to(dev).eval()
        self.evaluator = "fused_f16" if isinstance(net, FusedNet) else "torch"
        cfg = V1RootMCTSConfig(num_simulations=max(1, int(mcts_simulations)), exploration_weight=1.0,
                               temperature=float(temperature), add_dirichlet_noise=False, sample_moves=bool(sample_moves))
        self.mcts = V1RootMCTS(model=net, config=cfg, device=dev)
        self.temperature = float(temperature)

    def select(self, state: GpuStateBatch, force_uniform: Optional[torch.Tensor] = None):
        temps = torch.full((state.batch_size,), self.temperature, dtype=torch.float32, device=state.device)
        out = self.mcts.search_batch(state, temperatures=temps, add_dirichlet_noise=False,
                                     force_uniform_random_mask=force_uniform)
        return out.chosen_action_codes, out.chosen_valid_mask, out.terminal_mask


def play_matches(challenger, opponent, num_games: int, device, *, opening_random_moves: int = 0,
                 max_game_plies: int = 512, seed: Optional[int] = None, record_moves: bool = False) -> EvaluationStats:
    """All `num_games` games at once on `device`; returns the challenger's W/L/D (`record_moves`: plus every game's
    sequence of 220-d action indices in `move_log`)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("eval arena needs a HIP device (no CPU path)")
    if seed is not None:
        torch.manual_seed(int(seed))
        torch.cuda.manual_seed(int(seed))
    n = int(num_games)
    states = GpuStateBatch.initial(dev, n)
    plies = torch.zeros((n,), dtype=torch.int64, device=dev)
    done = torch.zeros((n,), dtype=torch.bool, device=dev)
    challenger_black = torch.arange(n, device=dev) < (n / 2)          # eval_checkpoint.py:487-495
    result_black = torch.zeros((n,), dtype=torch.float32, device=dev)
    finished = torch.zeros((n,), dtype=torch.bool, device=dev)
    log = []
    while True:
        active = torch.nonzero(~done).view(-1)
        if int(active.numel()) == 0:
            break
        sub = states.select(active)
        black_to_move = sub.current_player > 0
        chall_to_move = black_to_move == challenger_black.index_select(0, active)
        codes = torch.full((int(active.numel()), 4), -1, dtype=torch.int32, device=dev)
        valid = torch.zeros((int(active.numel()),), dtype=torch.bool, device=dev)
        term = torch.zeros_like(valid)
        opening = sub.move_count < int(opening_random_moves)
        for agent, who in ((challenger, chall_to_move), (opponent, ~chall_to_move)):
            rows = torch.nonzero(who).view(-1)
            if int(rows.numel()) == 0:
                continue
            part = sub.select(rows)
            force = opening.index_select(0, rows)
            if bool(force.all()):
                c, v, t = _uniform_legal_codes(part)
            else:
                c, v, t = agent.select(part, force if bool(force.any()) else None)
            codes.index_copy_(0, rows, c.to(torch.int32))
            valid.index_copy_(0, rows, v.to(torch.bool))
            term.index_copy_(0, rows, t.to(torch.bool))
        if record_moves:
            row = torch.full((n,), -1, dtype=torch.int32, device=dev)
            row.index_copy_(0, active, torch.where(valid & ~term, codes_to_indices(codes), torch.full_like(codes[:, 0], -1)))
            log.append(row)
        fin, res, _soft = v0_core.self_play_step_inplace(*states.tensors(), plies, done, active, codes, term, valid,
                                                         int(max_game_plies), 2.0)
        if int(fin.numel()) > 0:
            result_black.index_copy_(0, fin, res)
            finished.index_fill_(0, fin, True)
    res = torch.where(challenger_black, result_black, -result_black)
    win, loss, draw = res > 0, res < 0, res == 0
    cb = {}
    for name, sel in (("black", challenger_black), ("white", ~challenger_black)):
        cb[name] = {"wins": int((win & sel).sum()), "losses": int((loss & sel).sum()), "draws": int((draw & sel).sum()),
                    "games": int(sel.sum())}
    return EvaluationStats(wins=int(win.sum()), losses=int(loss.sum()), draws=int(draw.sum()), total_games=n,
                           color_breakdown=cb, move_log=torch.stack(log, dim=1) if (record_moves and log) else None)


def load_checkpoint_model(path: str):
    """Checkpoint (raw state_dict or {"model_state_dict": ...}) -> ChessNet of the matching architecture."""
    from .self_play_worker import _infer_model
    obj = torch.load(path, map_location="cpu", weights_only=False)
    state = obj["model_state_dict"] if isinstance(obj, dict) and "model_state_dict" in obj else obj
    model = _infer_model(state)
    model.load_state_dict(state, strict=True)
    return model.eval()


def evaluate_checkpoint(challenger_checkpoint: str, opponent_checkpoint: Optional[str] = None, *, num_games: int = 200,
                        device: str = "cuda:0", mcts_simulations: int = 64, temperature: float = 0.1,
                        sample_moves: bool = False, opening_random_moves: int = 0, max_game_plies: int = 512,
                        seed: int = 0) -> Dict[str, Any]:
    """vs-previous (two checkpoints) or vs-random (opponent None) probe; payload as eval_checkpoint.py:139-154."""
    games = int(num_games) if int(num_games) % 2 == 0 else max(2, (int(num_games) // 2) * 2)   # even (:48-54)
    chall = RootSearchAgent(load_checkpoint_model(challenger_checkpoint), device, mcts_simulations, temperature,
                            sample_moves)
    opp = RandomAgent() if not opponent_checkpoint else RootSearchAgent(
        load_checkpoint_model(opponent_checkpoint), device, mcts_simulations, temperature, sample_moves)
    stats = play_matches(chall, opp, games, device, opening_random_moves=opening_random_moves,
                         max_game_plies=max_game_plies, seed=seed)
    payload = stats.to_payload("vs_previous" if opponent_checkpoint else "vs_random")
    payload["seed"] = int(seed)
    return payload
