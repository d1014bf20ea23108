"""Steady-state self-play population: B concurrent games that are re-seated from the empty board the
moment they finish, so every step advances exactly B positions (used by bench.py and the worker's
throughput mode).  The per-step work is the reference's wave-loop body
(v1/python/self_play_gpu_runner.py:167-256): search -> trajectory append -> step -> finalize."""
from __future__ import annotations

from typing import Optional

import torch

from . import v0_core
from .mcts_gpu import GpuStateBatch, TOTAL_ACTION_DIM, V1RootMCTS, V1RootMCTSConfig
from .trajectory_buffer import TensorTrajectoryBuffer


class SteadyStateRootSelfPlay:
    def __init__(self, model, num_games: int, config: V1RootMCTSConfig, device, *, temperature_init: float = 1.0,
                 temperature_final: float = 0.1, temperature_threshold: int = 10, max_game_plies: int = 512,
                 arena_rows: Optional[int] = None, seed: int = 12345, fused_search: bool = True,
                 device_tail: bool = True, dual_stream: Optional[bool] = None) -> None:
        self.dev = torch.device(device)
        self.B = int(num_games)
        self.cfg = config
        self.mcts = V1RootMCTS(model, config, self.dev)
        # fixed population + fused network: the whole search as one captured, sync-free launch sequence
        self.fused = None
        if fused_search and hasattr(model, "desc"):
            from .root_search_fused import DualStreamRootSearch, FusedRootSearch
            # dual_stream: two halves on two streams, so that the bandit / prepare kernels of one half overlap the network
            # launches of the other -- measured 4 % SLOWER than one stream at C2 (two graphs, twice the small launches), so
            # it is off unless asked for
            dual = False if dual_stream is None else bool(dual_stream)
            self.dual_stream = bool(dual and self.B >= 2)
            self.fused = (DualStreamRootSearch if self.dual_stream else FusedRootSearch)(
                model, self.B, config.num_simulations, self.dev,
                                         exploration_weight=config.exploration_weight,
                                         add_dirichlet_noise=config.add_dirichlet_noise,
                                         dirichlet_alpha=config.dirichlet_alpha,
                                         dirichlet_epsilon=config.dirichlet_epsilon, sample_moves=config.sample_moves,
                                         soft_value_k=config.soft_value_k, seed=int(seed),
                                         sparse_ply=int(config.sparse_ply), sparse_top_k=int(config.sparse_top_k),
                                         child_eval_mode=str(config.child_eval_mode))
        self.t_init, self.t_final, self.t_thr = float(temperature_init), float(temperature_final), int(temperature_threshold)
        self.max_plies = int(max_game_plies)
        self.states = GpuStateBatch.initial(self.dev, self.B)
        self.plies = torch.zeros((self.B,), dtype=torch.int64, device=self.dev)
        self.done = torch.zeros((self.B,), dtype=torch.bool, device=self.dev)
        self.step_index = torch.full((self.B, self.max_plies), -1, dtype=torch.int64, device=self.dev)
        self.step_counts = torch.zeros((self.B,), dtype=torch.int64, device=self.dev)
        self.all_idx = torch.arange(self.B, dtype=torch.int64, device=self.dev)
        self.ones = torch.ones((self.B,), dtype=torch.int64, device=self.dev)
        self.buffer = TensorTrajectoryBuffer(self.dev, TOTAL_ACTION_DIM, initial_capacity=arena_rows or self.B * 64)
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        self._games_finished = 0
        self._reseated = torch.zeros((self.B,), dtype=torch.uint8, device=self.dev)   # slots re-seated by the last step
        self.positions = 0
        # device_tail: record / move / finalise / re-seat on the device (wave_tail.WaveTail), no host round trip per step
        self.tail = None
        if device_tail:
            from .wave_tail import WaveTail
            self.tail = WaveTail(self.buffer, self.B, self.max_plies, self.dev, soft_value_k=float(config.soft_value_k),
                                 reseat=True)
        self.outcome = self.tail.outcome if self.tail is not None else torch.zeros((3,), dtype=torch.int64, device=self.dev)

    @property
    def games_finished(self) -> int:
        return self._games_finished + (int(self.tail.finished.item()) if self.tail is not None else 0)

    @property
    def leaf_evals(self) -> int:
        return self.fused.leaf_evals if self.fused is not None else int(self.mcts._leaf_evals)

    def _reset_slots(self, slots: torch.Tensor) -> None:
        s = self.states
        s.board.index_fill_(0, slots, 0); s.marks_black.index_fill_(0, slots, False); s.marks_white.index_fill_(0, slots, False)
        s.phase.index_fill_(0, slots, 1); s.current_player.index_fill_(0, slots, 1)
        for t in (s.pending_marks_required, s.pending_marks_remaining, s.pending_captures_required,
                  s.pending_captures_remaining, s.forced_removals_done, s.move_count, s.moves_since_capture,
                  self.plies, self.step_counts):
            t.index_fill_(0, slots, 0)
        self.done.index_fill_(0, slots, False)

    def preroll(self, max_random_moves: int = 120) -> None:
        """Stagger the population: game g plays `target[g]` uniformly random legal moves (no search)."""
        target = torch.randint(0, max_random_moves + 1, (self.B,), generator=self.gen, device=self.dev)
        for t in range(max_random_moves):
            live = torch.nonzero(target > t).view(-1)
            if int(live.numel()) == 0:
                break
            sub = self.states.select(live)
            mask, meta = v0_core.encode_actions_fast(*sub.tensors()[:10], 36, 144, 36, 4)
            has = mask.any(dim=1)
            w = mask.to(torch.float32)
            w[~has, 0] = 1.0
            pick = torch.multinomial(w, 1, generator=self.gen).view(-1)
            codes = meta.gather(1, pick.view(-1, 1, 1).expand(-1, 1, 4)).view(-1, 4)
            fin, _, _ = v0_core.self_play_step_inplace(*self.states.tensors(), self.plies, self.done, live, codes,
                                                       ~has, has, self.max_plies, float(self.cfg.soft_value_k))
            if int(fin.numel()) > 0:
                self._reset_slots(fin)
        self.step_counts.zero_()

    def prepare(self) -> None:
        """Kernel loading and graph capture before anything is timed (one search whose result is discarded)."""
        temps = torch.ones((self.B,), dtype=torch.float32, device=self.dev)
        (self.fused or self.mcts).search_batch(self.states, temperatures=temps)
        torch.cuda.synchronize(self.dev)

    def step(self) -> None:
        temps = torch.where(self.plies < self.t_thr, self.t_init, self.t_final).to(torch.float32)
        if self.fused is not None:       # per-game RNG keys: the slot's game generation and the game's ply
            search = self.fused.search_batch(self.states, temperatures=temps, reset=self._reseated, rng_plies=self.plies)
            self._reseated.zero_()
            self.finish_step(search, reseated=self._reseated)
            return
        search = self.mcts.search_batch(self.states, temperatures=temps)
        self.finish_step(search)

    def finish_step(self, search, reseated: Optional[torch.Tensor] = None) -> None:
        """Trajectory rows, the chosen move, finalisation and re-seating of finished games for one searched ply."""
        if self.tail is not None:
            self.tail.record(self.states, self.done, self.step_index, self.step_counts, search)
            self.tail.step_finish(self.states, self.plies, self.done, self.step_index, self.step_counts, search,
                                  reseated=reseated)
            self.positions += self.B
            return
        rows = self.buffer.append_steps(search.model_input, search.legal_mask, search.policy_dense,
                                        self.states.current_player)
        self.step_index[self.all_idx, self.step_counts] = rows
        self.step_counts.add_(self.ones)
        fin, result, soft = v0_core.self_play_step_inplace(*self.states.tensors(), self.plies, self.done, self.all_idx,
                                                           search.chosen_action_codes, search.terminal_mask,
                                                           search.chosen_valid_mask, self.max_plies,
                                                           float(self.cfg.soft_value_k))
        self.positions += self.B
        if int(fin.numel()) > 0:
            _, _, out = self.buffer.finalize_games_inplace(step_index_matrix=self.step_index,
                                                           step_counts=self.step_counts, slots=fin,
                                                           result_from_black=result, soft_value_from_black=soft)
            self.outcome.add_(out)
            self._games_finished += int(fin.numel())
            self._reset_slots(fin)
            if reseated is not None:
                reseated.index_fill_(0, fin, 1)
