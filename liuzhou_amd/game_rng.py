"""Per-game counter RNG of the search engines (csrc/lz_rng.h: Philox4x32-10 keyed by the run seed, counter =
(game id, ply, purpose, index)).

Replaces the reference's draws from the device's global generator -- `torch.distributions.Gamma` for the Dirichlet
root noise and `torch.multinomial` for the move (v1/python/mcts_gpu.py:1329-1339,1410-1424), torch Dirichlet in
v1/python/portable_mcts.py:302-317 -- whose stream depends on the slot a game sits in and on the batch it is searched
with.  Here a game's noise and pick uniforms depend only on (seed, game id, ply): the same games are played
whichever slot, batch split, stream or rank runs them.  Parity runs still inject noise / uniforms explicitly.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

PURPOSE_NOISE, PURPOSE_PICK, PURPOSE_OPENING = 0, 1, 2


class GameRng:
    """Keys of B slots: `game` int64[B] (global game id of the slot's current game) and `ply` int64[B]."""

    def __init__(self, num_slots: int, device, seed: int = 12345, game_offset: int = 0,
                 game_stride: Optional[int] = None) -> None:
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("GameRng needs a HIP device (no CPU path)")
        self.B, self.device, self.seed = int(num_slots), dev, int(seed) & 0xFFFFFFFFFFFFFFFF
        self.stride = int(game_stride) if game_stride is not None else self.B
        self.game = torch.arange(self.B, dtype=torch.int64, device=dev) + int(game_offset)
        self.ply = torch.zeros((self.B,), dtype=torch.int64, device=dev)

    def begin_move(self, reset: Optional[torch.Tensor] = None, game_ids: Optional[torch.Tensor] = None,
                   plies: Optional[torch.Tensor] = None) -> None:
        """Keys of the move about to be searched.  With explicit `game_ids` / `plies` (the runner's own bookkeeping:
        WaveTail.slot_game, the `plies` tensor) those are used; otherwise slots flagged in `reset` start their next game
        (id += stride, so ids stay unique per slot) at ply 0."""
        if game_ids is not None:
            self.game.copy_(game_ids.to(torch.int64))
        elif reset is not None:
            self.game.add_(reset.to(torch.int64) * self.stride)
        if plies is not None:
            self.ply.copy_(plies.to(torch.int64))
        elif reset is not None:
            self.ply.mul_(1 - reset.to(torch.int64))

    def end_move(self, explicit_plies: bool = False) -> None:
        if not explicit_plies:
            self.ply.add_(1)

    def gamma_into(self, out: torch.Tensor, alpha: float, count: int) -> None:
        """out[g, k] = Gamma(alpha) draw k < count of slot g's (game, ply); normalised over the legal children by the
        expand kernel these are the Dirichlet(alpha) root noise."""
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_rng_gamma(C.c_uint64(self.seed), L.ptr(self.game), L.ptr(self.ply), L.i64(self.B),
                                         C.c_float(float(alpha)), L.i64(count), L.ptr(out), L.i64(int(out.stride(0))),
                                         L.stream_ptr(self.device)), "rng_gamma")

    def uniform_into(self, out: torch.Tensor, purpose: int = PURPOSE_PICK) -> None:
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_rng_uniform(C.c_uint64(self.seed), L.ptr(self.game), L.ptr(self.ply), L.i64(self.B),
                                           C.c_int(int(purpose)), L.ptr(out), L.stream_ptr(self.device)), "rng_uniform")
