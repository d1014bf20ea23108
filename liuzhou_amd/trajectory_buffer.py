"""Device-resident trajectory arena (mirror of v1/python/trajectory_buffer.py:11-211).

One row per recorded ply: model input f32[11,6,6], legal mask bool[220], policy target f32[220], value /
soft-value targets (NaN until the game ends) and the mover's sign.  2 692 B per sample; capacity is
reserved up front from (max plies x concurrent games) and doubles when exceeded.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import v0_core


@dataclass
class TensorSelfPlayBatch:
    state_tensors: torch.Tensor
    legal_masks: torch.Tensor
    policy_targets: torch.Tensor
    value_targets: torch.Tensor
    soft_value_targets: torch.Tensor

    @property
    def num_samples(self) -> int:
        return int(self.state_tensors.shape[0])

    def to(self, device) -> "TensorSelfPlayBatch":
        dev = torch.device(device)
        return TensorSelfPlayBatch(*(t.to(dev) for t in (self.state_tensors, self.legal_masks, self.policy_targets,
                                                          self.value_targets, self.soft_value_targets)))


class TensorTrajectoryBuffer:
    def __init__(self, device, action_dim: int, *, max_steps_hint: int = 512, concurrent_games_hint: int = 8,
                 initial_capacity: Optional[int] = None) -> None:
        self.device = torch.device(device)
        self.action_dim = int(action_dim)
        self._capacity = int(max(initial_capacity or 0, max(1, int(max_steps_hint) * int(concurrent_games_hint))))
        self._size = 0
        self._shape: Optional[Tuple[int, int, int]] = None
        self._state = self._legal = self._policy = self._value = self._soft = self._sign = None
        # device-side appends (wave_tail.WaveTail): the row cursor lives on the device, the host only tracks an upper
        # bound of it and reads the true value back when it has to (growth, host appends, build)
        self._cursor: Optional[torch.Tensor] = None
        self._upper = 0
        self._device_dirty = False

    def _allocate(self, capacity: int, shape: Tuple[int, int, int]) -> None:
        dev = self.device
        self._shape = shape
        self._state = torch.empty((capacity, *shape), dtype=torch.float32, device=dev)
        self._legal = torch.empty((capacity, self.action_dim), dtype=torch.bool, device=dev)
        self._policy = torch.empty((capacity, self.action_dim), dtype=torch.float32, device=dev)
        self._value = torch.full((capacity,), float("nan"), dtype=torch.float32, device=dev)
        self._soft = torch.full((capacity,), float("nan"), dtype=torch.float32, device=dev)
        self._sign = torch.empty((capacity,), dtype=torch.int8, device=dev)
        self._capacity = int(capacity)

    def _grow(self, required: int) -> None:
        old = (self._state, self._legal, self._policy, self._value, self._soft, self._sign)
        n = self._size
        self._allocate(max(int(required), 2 * max(1, self._capacity)), self._shape)
        for dst, src in zip((self._state, self._legal, self._policy, self._value, self._soft, self._sign), old):
            dst[:n].copy_(src[:n])

    # ---- device-side appends --------------------------------------------------------------------
    def reserve_rows(self, n: int, shape: Tuple[int, int, int] = (11, 6, 6)) -> torch.Tensor:
        """Make room for up to `n` more rows written by the device and return the int64[1] device cursor."""
        if self._state is None:
            self._allocate(self._capacity, tuple(shape))
        if self._cursor is None:
            self._cursor = torch.full((1,), self._size, dtype=torch.int64, device=self.device)
            self._upper = self._size
        if self._upper + int(n) > self._capacity:
            self.sync_cursor()
            if self._size + int(n) > self._capacity:
                self._grow(self._size + int(n))
        self._upper += int(n)
        self._device_dirty = True
        return self._cursor

    def reserve_slot_major(self, num_slots: int, max_steps: int, shape: Tuple[int, int, int] = (11, 6, 6)) -> None:
        """Slot-major live arena of the finished-row log (finished_log.py): row = slot * max_steps + step; the rows of a
        game leave for the log when the game ends, so `build()` has nothing to return in this mode."""
        need = int(num_slots) * int(max_steps)
        if self._state is None or self._capacity < need:
            if self._size or self._device_dirty:
                raise RuntimeError("TensorTrajectoryBuffer: slot-major mode needs an empty buffer")
            self._allocate(need, tuple(shape))

    def sync_cursor(self) -> int:
        """Read the device cursor back (one host synchronisation)."""
        if self._device_dirty:
            self._size = int(self._cursor.item())
            self._upper = self._size
            self._device_dirty = False
        return self._size

    @property
    def capacity(self) -> int:
        return self._capacity

    def arena(self):
        return self._state, self._legal, self._policy, self._value, self._soft, self._sign

    def append_step(self, model_input, legal_mask, policy_dense, player_sign: int) -> int:
        idx = self.append_steps(model_input.unsqueeze(0), legal_mask.unsqueeze(0), policy_dense.unsqueeze(0),
                                torch.tensor([int(player_sign)], dtype=torch.int64, device=model_input.device))
        return int(idx[0].item())

    def append_steps(self, model_input, legal_mask, policy_dense, player_sign) -> torch.Tensor:
        if model_input.dim() != 4:
            raise ValueError(f"model_input must be (N,C,H,W), got shape {tuple(model_input.shape)}")
        n = int(model_input.shape[0])
        for name, t in (("legal_mask", legal_mask), ("policy_dense", policy_dense)):
            if t.dim() != 2 or int(t.shape[0]) != n or int(t.shape[1]) != self.action_dim:
                raise ValueError(f"{name} must be (N,{self.action_dim}), got shape {tuple(t.shape)}")
        sign = torch.as_tensor(player_sign, device=model_input.device).view(-1)
        if int(sign.numel()) != n:
            raise ValueError(f"player_sign must have {n} elements, got {int(sign.numel())}")
        shape = tuple(int(x) for x in model_input.shape[1:])
        self.sync_cursor()
        if self._state is None:
            self._allocate(self._capacity, shape)
        elif self._shape != shape:
            raise ValueError(f"Inconsistent state shape: expected {self._shape}, got {shape}")
        end = self._size + n
        if end > self._capacity:
            self._grow(end)
        s = slice(self._size, end)
        self._state[s].copy_(model_input.detach())
        self._legal[s].copy_(legal_mask.detach())
        self._policy[s].copy_(policy_dense.detach())
        self._value[s].fill_(float("nan"))
        self._soft[s].fill_(float("nan"))
        self._sign[s].copy_(torch.where(sign >= 0, 1, -1).to(torch.int8))
        start, self._size = self._size, end
        if self._cursor is not None:
            self._cursor.fill_(end)
            self._upper = end
        return torch.arange(start, end, dtype=torch.int64, device=self.device)

    def finalize_games_inplace(self, *, step_index_matrix, step_counts, slots, result_from_black,
                               soft_value_from_black):
        self.sync_cursor()
        if self._size == 0:
            e = torch.empty((0,), dtype=torch.int64, device=self.device)
            return e, e.clone(), torch.zeros((3,), dtype=torch.int64, device=self.device)
        return v0_core.finalize_trajectory_inplace(self._value, self._soft, self._sign, step_index_matrix,
                                                   step_counts, slots, result_from_black, soft_value_from_black)

    def build(self) -> TensorSelfPlayBatch:
        self.sync_cursor()
        if self._size == 0:
            shape = self._shape or (11, 6, 6)
            dev = self.device
            return TensorSelfPlayBatch(
                torch.empty((0, *shape), dtype=torch.float32, device=dev),
                torch.empty((0, self.action_dim), dtype=torch.bool, device=dev),
                torch.empty((0, self.action_dim), dtype=torch.float32, device=dev),
                torch.empty((0,), dtype=torch.float32, device=dev), torch.empty((0,), dtype=torch.float32, device=dev))
        n = self._size
        return TensorSelfPlayBatch(self._state[:n].clone(), self._legal[:n].clone(), self._policy[:n].clone(),
                                   self._value[:n].clone(), self._soft[:n].clone())
