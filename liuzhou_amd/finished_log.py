"""Finished-row log: the device half of the streaming self-play worker.

The reference worker (v1/python/self_play_worker.py:430-546) plays a chunk of games to the end, then copies the whole
chunk to the host and writes it with the GPU idle, chunk after chunk.  Here ONE wave plays the whole shard (finished
slots start the next game at once) and the rows of a game leave the device the moment the game has ended:

* the live rows sit in a SLOT-MAJOR arena (`row = slot * max_steps + step`, `lz_wave_record` with a NULL cursor);
* after every ply `lz_wave_log_finished` appends the rows of the games that have just ended to a GAME-MAJOR log (ordered
  scan: the log order is deterministic) and frees their slots;
* the host cuts the log into SEGMENTS (about `segment_games` finished games each) by switching to another log arena and
  hands the full one to a consumer (`on_segment`), which copies it out on its own stream and thread while the wave goes
  on playing, and gives the arena back (`Segment.release`).

The host never waits for the device here: it reads the log counters of two plies ago from pinned memory (the same
latency as the wave loop's `all done` flag) and the kernels apply back-pressure by themselves -- a game whose rows do not
fit the current arena keeps its slot until the host has switched arenas (`lz_wave_reseat(logged_only=1)`).
"""
from __future__ import annotations

import queue
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch

from . import _lib as L


@dataclass
class LogArena:
    state: torch.Tensor          # f32[cap, 11, 6, 6]
    legal: torch.Tensor          # bool[cap, A]
    policy: torch.Tensor         # f32[cap, A]
    value: torch.Tensor          # f32[cap]
    soft: torch.Tensor           # f32[cap]
    counters: torch.Tensor       # i64[4]: rows, games, games waiting, rows waiting
    index: int = 0


@dataclass
class Segment:
    """One cut of the log.  `ready` is recorded on the playing stream behind the last kernel that wrote the arena;
    the exact row / game counts are `arena.counters[:2]` once `ready` has passed.  Call `release()` when the arena's
    contents have been copied out."""
    arena: LogArena
    ready: torch.cuda.Event
    number: int
    final: bool
    _free: "queue.Queue[int]" = field(repr=False, default=None)

    def release(self) -> None:
        self._free.put(self.arena.index)


class FinishedRowLog:
    def __init__(self, device, *, segment_games: int, num_slots: int, max_steps: int, action_dim: int = 220,
                 capacity_rows: Optional[int] = None, num_arenas: int = 2,
                 on_segment: Optional[Callable[[Segment], None]] = None) -> None:
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("FinishedRowLog needs a HIP device (no CPU path)")
        self.device, self.G, self.Tmax, self.A = dev, int(num_slots), int(max_steps), int(action_dim)
        self.segment_games = max(1, int(segment_games))
        # a segment of `segment_games` games of ~130 rows, plus the games that end while the host is two plies behind
        # (~G / 130 per ply); whatever does not fit waits a ply or two in its slot (back-pressure), nothing is lost
        cap = int(capacity_rows) if capacity_rows else self.segment_games * 160 + 2 * self.G
        self.capacity = max(cap, 2 * self.Tmax)
        self.on_segment = on_segment
        self.on_blocked: Optional[Callable[[], None]] = None      # called when a cut finds no free arena (consumer health check)
        self.arenas: List[LogArena] = []
        for i in range(max(2, int(num_arenas))):
            self.arenas.append(LogArena(
                torch.empty((self.capacity, 11, 6, 6), dtype=torch.float32, device=dev),
                torch.empty((self.capacity, self.A), dtype=torch.bool, device=dev),
                torch.empty((self.capacity, self.A), dtype=torch.float32, device=dev),
                torch.empty((self.capacity,), dtype=torch.float32, device=dev),
                torch.empty((self.capacity,), dtype=torch.float32, device=dev),
                torch.zeros((4,), dtype=torch.int64, device=dev), i))
        self._free: "queue.Queue[int]" = queue.Queue()
        for a in self.arenas[1:]:
            self._free.put(a.index)
        self.cur = self.arenas[0]
        self.gen = 0                                            # bumped at every switch: stale read-backs are ignored
        self.base = torch.empty((self.G,), dtype=torch.int64, device=dev)
        self._pinned = [torch.zeros((4,), dtype=torch.int64).pin_memory() for _ in range(2)]
        self._pinned_gen = [-1, -1]
        self.segments_cut = 0
        self.blocked_polls = 0                                  # polls that wanted to cut and found no free arena
        self.live = None

    # ---- per ply (called by WaveTail.run on the playing stream) -----------------------------------------------------
    def bind(self, live_arena) -> None:
        """`live_arena` = (state, legal, policy, value, soft, sign) of the slot-major TensorTrajectoryBuffer."""
        self.live = live_arena

    def after_ply(self, done: torch.Tensor, step_counts: torch.Tensor, k: int) -> None:
        a, (s, lg, p, v, sf, _sign) = self.cur, self.live
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_wave_log_finished(
                L.ptr(done), L.ptr(step_counts), L.i64(self.G), L.i64(self.Tmax), L.i64(self.A), L.ptr(s), L.ptr(lg),
                L.ptr(p), L.ptr(v), L.ptr(sf), L.ptr(a.state), L.ptr(a.legal), L.ptr(a.policy), L.ptr(a.value),
                L.ptr(a.soft), L.i64(self.capacity), L.ptr(a.counters), L.ptr(self.base),
                L.stream_ptr(self.device)), "wave_log_finished")
        self._pinned[k].copy_(a.counters, non_blocking=True)
        self._pinned_gen[k] = self.gen

    def waiting(self) -> torch.Tensor:
        """Device scalar: games whose rows did not fit the current arena (they hold their slots)."""
        return self.cur.counters[2]

    def poll(self, k: int) -> None:
        """Look at the counters of two plies ago (the caller has synchronised the event that covers the copy)."""
        if self._pinned_gen[k] != self.gen:
            return
        rows, games, waiting, _ = (int(x) for x in self._pinned[k].tolist())
        if games >= self.segment_games or waiting > 0 or rows >= self.capacity - self.G:
            if not self._switch(final=False):
                self.blocked_polls += 1
                if self.on_blocked is not None:
                    self.on_blocked()

    def _switch(self, final: bool) -> bool:
        nxt = None
        if not final:
            try:
                nxt = self._free.get_nowait()
            except queue.Empty:
                return False
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        seg = Segment(self.cur, ev, self.segments_cut, final, self._free)
        self.segments_cut += 1
        self.gen += 1
        if nxt is not None:
            self.cur = self.arenas[nxt]
            self.cur.counters.zero_()                            # on the playing stream, before the next log kernel
        if self.on_segment is not None:
            self.on_segment(seg)
        else:
            seg.release()
        return True

    def close(self) -> None:
        """Hand over what is left (the run has ended: every finished game is in a log)."""
        self._switch(final=True)
