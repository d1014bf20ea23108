"""Device-side tail of one ply of the v1 wave loop for a fixed wave of slots: trajectory rows, the move and the
finalisation of finished games without a host round trip (`lz_wave_record`, `lz_wave_step_finish`).

It replaces, for whole waves, the sequence `append_steps` -> `step_index[...] = rows` -> `self_play_step_inplace` ->
`finalize_games_inplace` of v1/python/self_play_gpu_runner.py:205-247, whose `nonzero`-shaped outputs force one host
synchronisation per ply; finished slots stay in the batch and are masked by `done`."""
from __future__ import annotations

import ctypes as C
import time
from typing import Optional

import torch

from . import _lib as L
from .mcts_gpu import GpuStateBatch, RootSearchBatchOutput
from .trajectory_buffer import TensorTrajectoryBuffer

DELTA_BINS = 37


class WaveTail:
    def __init__(self, buffer: TensorTrajectoryBuffer, num_slots: int, max_game_plies: int, device,
                 soft_value_k: float = 2.0, reseat: bool = False, row_log=None) -> None:
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("WaveTail needs a HIP device (no CPU path)")
        self.buffer, self.G, self.max_plies, self.device = buffer, int(num_slots), int(max_game_plies), dev
        self.soft_k, self.reseat = float(soft_value_k), bool(reseat)
        z = lambda n, dt: torch.zeros((n,), dtype=dt, device=dev)
        self.rows = z(self.G, torch.int64)
        self.overflow = z(1, torch.int32)
        self.outcome = z(3, torch.int64)
        self.delta_hist = z(DELTA_BINS, torch.int64)
        self.finished = z(1, torch.int64)
        self.slot_game = torch.arange(self.G, dtype=torch.int64, device=dev)   # game number played in each slot (run())
        self.collect_timing = False
        self._timing_events = []
        self.live_estimate = self.G                          # run(): live slots two plies ago (host-visible without a wait)
        self.host_wait_ms = self.loop_ms = 0.0              # run(): time the host spent waiting for the device / in the loop
        self.plies_launched = 0
        # finished_log.FinishedRowLog: the live rows are slot-major (row = slot * max_plies + step, no step_index matrix)
        # and the rows of a game move to the log when the game ends (the streaming worker)
        self.row_log = row_log
        if row_log is not None:
            if reseat:
                raise ValueError("WaveTail: the finished-row log needs run()'s re-seating (reseat=False)")
            if int(row_log.G) != self.G or int(row_log.Tmax) != self.max_plies:
                raise ValueError("WaveTail: the finished-row log was built for another wave shape")

    def _bracket(self, name: str):
        """HIP-event bracket + roctx range for one tail kernel sequence (bucket names of the reference's runner,
        v1/python/self_play_gpu_runner.py:276-281)."""
        from contextlib import contextmanager

        @contextmanager
        def cm():
            torch.cuda.nvtx.range_push(f"lz.tail.{name}")
            if self.collect_timing:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(torch.cuda.current_stream(self.device))
            try:
                yield
            finally:
                if self.collect_timing:
                    b.record(torch.cuda.current_stream(self.device))
                    self._timing_events.append((name, a, b))
                torch.cuda.nvtx.range_pop()
        return cm()

    def get_timing(self):
        torch.cuda.synchronize(self.device)
        out = {}
        for name, a, b in self._timing_events:
            d = out.setdefault(name, {"ms": 0.0, "calls": 0})
            d["ms"] += float(a.elapsed_time(b)); d["calls"] += 1
        self._timing_events = []
        return out

    def record(self, states: GpuStateBatch, done: torch.Tensor, step_index: torch.Tensor, step_counts: torch.Tensor,
               search: RootSearchBatchOutput) -> None:
        """Append this ply's sample of every live slot to the trajectory arena."""
        G = self.G
        shape = tuple(int(x) for x in search.model_input.shape[1:])
        if self.row_log is not None:
            cursor, steps = None, self.max_plies
            self.buffer.reserve_slot_major(G, steps, shape)
        else:
            cursor, steps = self.buffer.reserve_rows(G, shape), int(step_index.shape[1])
        a_state, a_legal, a_policy, a_value, a_soft, a_sign = self.buffer.arena()
        mi, lm, pol = search.model_input.contiguous(), search.legal_mask.contiguous(), search.policy_dense.contiguous()
        T = int(pol.shape[1])
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_wave_record(
                L.ptr(done), L.i64(G), L.ptr(cursor), L.i64(self.buffer.capacity), L.i64(steps),
                L.ptr(step_index), L.ptr(step_counts), L.ptr(self.rows), L.ptr(self.overflow), L.ptr(mi), L.ptr(lm),
                L.ptr(pol), L.ptr(states.current_player), L.i64(T), L.ptr(a_state), L.ptr(a_legal), L.ptr(a_policy),
                L.ptr(a_value), L.ptr(a_soft), L.ptr(a_sign), L.stream_ptr(self.device)), "wave_record")

    def step_finish(self, states: GpuStateBatch, plies: torch.Tensor, done: torch.Tensor, step_index: torch.Tensor,
                    step_counts: torch.Tensor, search: RootSearchBatchOutput, lengths: Optional[torch.Tensor] = None,
                    reseated: Optional[torch.Tensor] = None, slot_game: Optional[torch.Tensor] = None) -> None:
        """Play the chosen move of every live slot; finalise (and re-seat, if asked) the games that end."""
        _, _, _, a_value, a_soft, a_sign = self.buffer.arena()
        ts = states.tensors()
        if any(not t.is_contiguous() for t in ts):
            raise RuntimeError("WaveTail.step_finish: state tensors must be contiguous (they are updated in place)")
        codes = search.chosen_action_codes.contiguous()
        term = search.terminal_mask.contiguous()
        cvalid = search.chosen_valid_mask.contiguous()
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_wave_step_finish(
                C.byref(L.soa(ts)), L.i64(self.G), L.ptr(plies), L.ptr(done), L.ptr(codes), L.ptr(term), L.ptr(cvalid),
                L.i64(self.max_plies), C.c_float(self.soft_k), L.ptr(a_value), L.ptr(a_soft), L.ptr(a_sign),
                L.ptr(step_index), L.ptr(step_counts),
                L.i64(self.max_plies if step_index is None else int(step_index.shape[1])), L.ptr(self.outcome),
                L.ptr(self.delta_hist), L.ptr(lengths), L.ptr(slot_game), L.ptr(self.finished), L.ptr(reseated),
                C.c_int(1 if self.reseat else 0), L.stream_ptr(self.device)), "wave_step_finish")

    def start_next_games(self, states: GpuStateBatch, plies: torch.Tensor, done: torch.Tensor, step_counts: torch.Tensor,
                         budget: torch.Tensor, next_game: torch.Tensor, slot_game: torch.Tensor,
                         reseated: Optional[torch.Tensor] = None) -> None:
        """Finished slots restart from the empty board (ascending slot order) while `budget` games remain to be started."""
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_wave_reseat(C.byref(L.soa(states.tensors())), L.i64(self.G), L.ptr(done), L.ptr(plies),
                                           L.ptr(step_counts), L.ptr(budget), L.ptr(next_game), L.ptr(slot_game),
                                           L.ptr(reseated), C.c_int(1 if self.row_log is not None else 0),
                                           L.stream_ptr(self.device)), "wave_reseat")

    def run(self, search_fn, states: GpuStateBatch, plies: torch.Tensor, done: torch.Tensor, step_index: torch.Tensor,
            step_counts: torch.Tensor, lengths: torch.Tensor, t_init: float, t_final: float, t_threshold: int,
            games_to_start: int = 0) -> int:
        """The wave loop with nothing per ply on the host: `search_fn(states, temperatures, done, reseated)` ->
        RootSearchBatchOutput for all slots (finished ones are masked by `done`; `reseated` uint8 marks slots that
        have just started a new game), then record / move / finalise on the device.  `games_to_start` > 0: a slot
        whose game has finished starts the next game at once (lz_wave_reseat) instead of idling until the whole wave
        is done; `lengths` is then indexed by game.  The loop ends on the `all done` flag of TWO plies ago (copied to
        pinned memory behind an event), so the host stays one ply ahead of the device; the price is exactly one
        extra, fully masked ply at the end.  Returns the number of plies launched (including that one).
        With a finished-row log (`row_log`) `step_index` is None, the rows of the games that have ended move to the log
        after every ply, and the loop ends when every game has ended AND is in a log (which the caller then closes)."""
        dev, g = self.device, self.G
        log = self.row_log
        if (log is None) == (step_index is None):
            raise ValueError("WaveTail.run: pass a step_index matrix, or build the tail with a finished-row log")
        flags = [torch.zeros((1,), dtype=torch.bool).pin_memory() for _ in range(2)]
        live = [torch.full((1,), g, dtype=torch.int64).pin_memory() for _ in range(2)]     # live slots, as of two plies ago
        self.live_estimate = g
        events = [torch.cuda.Event() for _ in range(2)]
        budget = torch.full((1,), int(games_to_start), dtype=torch.int64, device=dev)
        next_game = torch.full((1,), g, dtype=torch.int64, device=dev)
        slot_game = self.slot_game
        slot_game.copy_(torch.arange(g, dtype=torch.int64, device=dev))
        reseated = torch.zeros((g,), dtype=torch.uint8, device=dev)
        ply = 0
        t_run = time.perf_counter()
        while True:
            k = ply & 1
            if ply >= 2:
                t_w = time.perf_counter()
                events[k].synchronize()
                self.host_wait_ms += (time.perf_counter() - t_w) * 1e3      # ~0 for a whole run: the HOST is the bottleneck
                if bool(flags[k].item()):
                    break
                self.live_estimate = int(live[k].item())          # a search may size its launches by it (compact lists)
                if log is not None:
                    log.poll(k)                                   # may switch log arenas (a segment leaves)
            if games_to_start > 0 and ply > 0:
                self.start_next_games(states, plies, done, step_counts, budget, next_game, slot_game, reseated)
            temps = torch.where(plies < int(t_threshold), float(t_init), float(t_final)).to(torch.float32)
            search = search_fn(states, temps, done, reseated)
            reseated.zero_()
            with self._bracket("finalize_ms"):
                self.record(states, done, step_index, step_counts, search)
            with self._bracket("self_play_step_ms"):
                self.step_finish(states, plies, done, step_index, step_counts, search, lengths=lengths, slot_game=slot_game)
            # all finished and nothing left to start (a finished slot restarts at the top of the next ply otherwise)
            if log is not None:
                if ply == 0:
                    log.bind(self.buffer.arena())
                log.after_ply(done, step_counts, k)
                flags[k].copy_((done.all() & (budget <= 0).all() & (log.waiting() == 0)).view(1), non_blocking=True)
            else:
                flags[k].copy_((done.all() & (budget <= 0).all()).view(1), non_blocking=True)
            live[k].copy_((~done).sum().view(1), non_blocking=True)
            events[k].record(torch.cuda.current_stream(dev))
            ply += 1
        self.loop_ms += (time.perf_counter() - t_run) * 1e3
        self.plies_launched += ply
        return ply

    def check_overflow(self) -> None:
        n = int(self.overflow.item())
        if n:
            raise RuntimeError(f"WaveTail: {n} trajectory rows were dropped (arena or step-index capacity too small)")
