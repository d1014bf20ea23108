"""Root-PUCT search over the HIP operator surface (host-side mirror of v1/python/mcts_gpu.py).

Same public interface as the reference module -- `GpuStateBatch`, `V1RootMCTSConfig`,
`V1RootMCTS(model, config, device, inference_engine=None, collect_timing=False)`,
`.search_batch(state, temperatures=, add_dirichlet_noise=, force_uniform_random_mask=)` returning a
`RootSearchBatchOutput` -- but every operator runs as a gfx950 kernel from libliuzhou_hip.so and the
per-ply work is organised to need a single host read (the number of child states) instead of the
reference's many `.item()` synchronisations.

Search semantics ("variant R", mcts_gpu.py:1249-1457): evaluate the root and all of its children once
(value only), then distribute `num_simulations` PUCT pulls over the fixed child values.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Dict, Optional, Tuple

import torch

from . import v0_core
from .net import bucket_logits_to_scalar

PLACEMENT_DIM, MOVEMENT_DIM, SELECTION_DIM, AUXILIARY_DIM = 36, 144, 36, 4   # v0/python/move_encoder.py:46-51
TOTAL_ACTION_DIM = PLACEMENT_DIM + MOVEMENT_DIM + SELECTION_DIM + AUXILIARY_DIM
MAX_MOVE_COUNT, NO_CAPTURE_DRAW_LIMIT, LOSE_PIECE_THRESHOLD = 144, 36, 4   # src/game_state.py:29-31
PHASE_MOVEMENT = int(v0_core.Phase.MOVEMENT)
PHASE_CAPTURE_SELECTION = int(v0_core.Phase.CAPTURE_SELECTION)
PHASE_COUNTER_REMOVAL = int(v0_core.Phase.COUNTER_REMOVAL)

_STATE_FIELDS = ("board", "marks_black", "marks_white", "phase", "current_player", "pending_marks_required",
                 "pending_marks_remaining", "pending_captures_required", "pending_captures_remaining",
                 "forced_removals_done", "move_count", "moves_since_capture")


@dataclass
class GpuStateBatch:
    """12-tensor state batch (mcts_gpu.py:40-145): board int8[B,6,6], marks bool[B,6,6] x2, 9 x int64[B]."""

    board: torch.Tensor
    marks_black: torch.Tensor
    marks_white: torch.Tensor
    phase: torch.Tensor
    current_player: torch.Tensor
    pending_marks_required: torch.Tensor
    pending_marks_remaining: torch.Tensor
    pending_captures_required: torch.Tensor
    pending_captures_remaining: torch.Tensor
    forced_removals_done: torch.Tensor
    move_count: torch.Tensor
    moves_since_capture: torch.Tensor

    @property
    def device(self) -> torch.device:
        return self.board.device

    @property
    def batch_size(self) -> int:
        return int(self.board.shape[0])

    def tensors(self) -> Tuple[torch.Tensor, ...]:
        return tuple(getattr(self, f) for f in _STATE_FIELDS)

    def _map(self, fn) -> "GpuStateBatch":
        return GpuStateBatch(*(fn(t) for t in self.tensors()))

    def to(self, device) -> "GpuStateBatch":
        dev = torch.device(device)
        return self._map(lambda t: t.to(dev))

    def slice(self, index: int) -> "GpuStateBatch":
        return self._map(lambda t: t[index:index + 1])

    def select(self, indices) -> "GpuStateBatch":
        if isinstance(indices, list):
            if not indices:
                raise ValueError("indices must not be empty.")
            idx = torch.tensor(indices, dtype=torch.int64, device=self.device)
        else:
            idx = indices.to(device=self.device, dtype=torch.int64).view(-1)
            if int(idx.numel()) == 0:
                raise ValueError("indices must not be empty.")
        return self._map(lambda t: t.index_select(0, idx))

    @staticmethod
    def initial(device, batch_size: int = 1) -> "GpuStateBatch":
        dev = torch.device(device)
        n = int(batch_size)
        z = lambda: torch.zeros((n,), dtype=torch.int64, device=dev)
        return GpuStateBatch(
            board=torch.zeros((n, 6, 6), dtype=torch.int8, device=dev),
            marks_black=torch.zeros((n, 6, 6), dtype=torch.bool, device=dev),
            marks_white=torch.zeros((n, 6, 6), dtype=torch.bool, device=dev),
            phase=torch.ones((n,), dtype=torch.int64, device=dev),
            current_player=torch.ones((n,), dtype=torch.int64, device=dev),
            pending_marks_required=z(), pending_marks_remaining=z(), pending_captures_required=z(),
            pending_captures_remaining=z(), forced_removals_done=z(), move_count=z(), moves_since_capture=z())


def states_to_model_input(batch: GpuStateBatch) -> torch.Tensor:
    return v0_core.states_to_model_input(batch.board, batch.marks_black, batch.marks_white, batch.phase,
                                         batch.current_player)


def encode_actions_fast(batch: GpuStateBatch) -> Tuple[torch.Tensor, torch.Tensor]:
    return v0_core.encode_actions_fast(*batch.tensors()[:10], PLACEMENT_DIM, MOVEMENT_DIM, SELECTION_DIM,
                                       AUXILIARY_DIM)


def batch_apply_moves_compat(batch: GpuStateBatch, action_codes: torch.Tensor,
                             parent_indices: torch.Tensor) -> GpuStateBatch:
    out = v0_core.batch_apply_moves(*batch.tensors(), action_codes, parent_indices)
    if len(out) != 12:
        raise RuntimeError(f"Unexpected batch_apply_moves output arity: {len(out)}")
    return GpuStateBatch(*out)


@dataclass
class V1RootMCTSConfig:
    """Defaults as mcts_gpu.py:223-236."""
    num_simulations: int = 128
    exploration_weight: float = 1.0
    temperature: float = 1.0
    add_dirichlet_noise: bool = True
    dirichlet_alpha: float = 0.3
    dirichlet_epsilon: float = 0.25
    sample_moves: bool = True
    autocast_dtype: str = "float16"
    child_eval_mode: str = "value_only"
    soft_value_k: float = 2.0
    sparse_ply: int = 1
    sparse_top_k: int = 8


@dataclass
class RootSearchOutput:
    model_input: torch.Tensor
    legal_mask: torch.Tensor
    policy_dense: torch.Tensor
    root_value: float
    terminal: bool
    chosen_action_index: Optional[int]
    chosen_action_code: Optional[torch.Tensor]


@dataclass
class RootSearchBatchOutput:
    model_input: torch.Tensor
    legal_mask: torch.Tensor
    policy_dense: torch.Tensor
    root_value: torch.Tensor
    terminal_mask: torch.Tensor
    chosen_action_indices: torch.Tensor
    chosen_action_codes: torch.Tensor
    chosen_valid_mask: torch.Tensor


class V1RootMCTS:
    """Root-only PUCT on the GPU.  `injected_noise` / `injected_uniforms` (set as attributes before a
    call) replace the on-device RNG draws for parity runs: noise is a [R, Amax] non-negative tensor that
    is normalised over legal actions, uniforms is [R] in [0,1)."""

    def __init__(self, model, config: V1RootMCTSConfig, device, inference_engine=None,
                 collect_timing: bool = False) -> None:
        mode = str(config.child_eval_mode).strip().lower()
        if mode not in ("value_only", "full"):
            raise ValueError(f"unknown child_eval_mode: {config.child_eval_mode}")
        self.model = model
        self.config = config
        self.device = torch.device(device)
        self.inference_engine = inference_engine
        self._child_eval_mode = mode
        self._collect_timing = bool(collect_timing)
        self._timing_ms: Dict[str, float] = {"root_puct_ms": 0.0, "pack_writeback_ms": 0.0}
        self._timing_calls: Dict[str, int] = {"root_puct_ms": 0, "pack_writeback_ms": 0}
        self._events = []
        self._terminal_soft_override_count = 0
        self._forced_uniform_pick_count = 0
        self._leaf_evals = 0
        self.injected_noise: Optional[torch.Tensor] = None
        self.injected_uniforms: Optional[torch.Tensor] = None

    # ---- timing (CUDA-event based like mcts_gpu.py:576-602, drained lazily) ----
    class _Timed:
        def __init__(self, owner, name):
            self.o, self.n = owner, name

        def __enter__(self):
            if self.o._collect_timing and self.o.device.type == "cuda":
                self.s = torch.cuda.Event(enable_timing=True); self.e = torch.cuda.Event(enable_timing=True)
                self.s.record()

        def __exit__(self, *a):
            if self.o._collect_timing and self.o.device.type == "cuda":
                self.e.record()
                self.o._events.append((self.n, self.s, self.e))

    def _timed(self, name):
        return V1RootMCTS._Timed(self, name)

    def get_timing(self, reset: bool = False) -> Dict[str, Dict]:
        if self._events:
            torch.cuda.synchronize(self.device)
            for name, s, e in self._events:
                self._timing_ms[name] += float(s.elapsed_time(e))
                self._timing_calls[name] += 1
            self._events = []
        out = {
            "timing_ms": dict(self._timing_ms),
            "timing_calls": dict(self._timing_calls),
            "counters": {
                "terminal_soft_override_count": int(self._terminal_soft_override_count),
                "forced_uniform_pick_count": int(self._forced_uniform_pick_count),
                "leaf_eval_count": int(self._leaf_evals),
                "finalize_graph_capture_count": 0, "finalize_graph_replay_count": 0,
                "finalize_graph_fallback_count": 0,
            },
        }
        if reset:
            for k in self._timing_ms:
                self._timing_ms[k] = 0.0
                self._timing_calls[k] = 0
        return out

    # ---- static helpers (mcts_gpu.py:658-708) ----
    @staticmethod
    def _terminal_mask_from_next_state(batch: GpuStateBatch) -> torch.Tensor:
        post = batch.phase.eq(PHASE_MOVEMENT) | batch.phase.eq(PHASE_CAPTURE_SELECTION) | batch.phase.eq(
            PHASE_COUNTER_REMOVAL)
        black = batch.board.eq(1).sum(dim=(1, 2))
        white = batch.board.eq(-1).sum(dim=(1, 2))
        win = post & (black.lt(LOSE_PIECE_THRESHOLD) | white.lt(LOSE_PIECE_THRESHOLD))
        draw = batch.move_count.ge(MAX_MOVE_COUNT) | batch.moves_since_capture.ge(NO_CAPTURE_DRAW_LIMIT)
        return win | draw

    @staticmethod
    def _soft_tanh_from_board_black(board: torch.Tensor, soft_value_k: float) -> torch.Tensor:
        black = board.eq(1).sum(dim=(1, 2)).to(torch.float32)
        white = board.eq(-1).sum(dim=(1, 2)).to(torch.float32)
        return torch.tanh(((black - white) / 18.0) * float(soft_value_k))

    @staticmethod
    def _child_values_to_parent_perspective(child_values, parent_players, child_players) -> torch.Tensor:
        vals = child_values.to(torch.float32).view(-1)
        par = parent_players.to(torch.int64).view(-1)
        chi = child_players.to(torch.int64).view(-1)
        if not (vals.numel() == par.numel() == chi.numel()):
            raise ValueError("child/parent perspective tensors must align: "
                             f"values={vals.numel()}, parents={par.numel()}, children={chi.numel()}")
        return torch.where(chi.eq(par), vals, -vals)

    # ---- network ----
    def _autocast(self):
        if self.device.type != "cuda":
            return torch.autocast("cpu", enabled=False)
        key = str(self.config.autocast_dtype).strip().lower()
        if key in ("fp32", "float32", "none", "off"):
            return torch.autocast("cuda", enabled=False)
        return torch.autocast("cuda", dtype=torch.bfloat16 if key in ("bf16", "bfloat16") else torch.float16)

    def _fused(self):
        """The fused gfx950 forward (net_hip.FusedNet) if one was given as `model` or `inference_engine`."""
        for cand in (self.inference_engine, self.model):
            if cand is not None and hasattr(cand, "values_only") and hasattr(cand, "last_value"):
                return cand
        return None

    def _forward_model(self, inputs: torch.Tensor):
        self._leaf_evals += int(inputs.shape[0])
        fused = self._fused()
        if fused is not None:
            return fused(inputs)
        if self.inference_engine is not None:
            return self.inference_engine.forward(inputs, int(inputs.shape[0]))
        self.model.eval()
        with torch.inference_mode():
            with self._autocast():
                return self.model(inputs)

    def _to_scalar_value(self, raw: torch.Tensor) -> torch.Tensor:
        if raw.dim() == 2 and raw.size(1) == 3:
            p = torch.softmax(raw.float(), dim=1)
            return p[:, 0] - p[:, 2]
        if raw.dim() == 2 and raw.size(1) == 1:
            return raw[:, 0].float()
        if raw.dim() == 2 and raw.size(1) >= 2:
            return bucket_logits_to_scalar(raw.float(), num_bins=int(raw.size(1)))
        return raw.view(-1).float()

    def _evaluate_batch(self, batch: GpuStateBatch):
        inputs = states_to_model_input(batch)
        lp1, lp2, lpm, raw = self._forward_model(inputs)
        fused = self._fused()
        values = fused.last_value if fused is not None else self._to_scalar_value(raw).float()
        legal_mask, metadata = encode_actions_fast(batch)
        probs, _ = v0_core.project_policy_logits_fast(lp1.float(), lp2.float(), lpm.float(), legal_mask,
                                                      PLACEMENT_DIM, MOVEMENT_DIM, SELECTION_DIM, AUXILIARY_DIM)
        return inputs, legal_mask, metadata, probs, values

    def _evaluate_values_only(self, batch: GpuStateBatch) -> torch.Tensor:
        fused = self._fused()
        if fused is not None:
            inputs = states_to_model_input(batch)
            self._leaf_evals += int(inputs.shape[0])
            return fused.values_only(inputs)
        _, _, _, raw = self._forward_model(states_to_model_input(batch))
        return self._to_scalar_value(raw).float()

    def apply_action(self, state: GpuStateBatch, action_code: torch.Tensor) -> GpuStateBatch:
        codes = action_code.view(1, 4) if action_code.dim() == 1 else action_code
        n = int(codes.shape[0])
        if state.batch_size == 1:
            parents = torch.zeros((n,), dtype=torch.int64, device=state.device)
        elif n == state.batch_size:
            parents = torch.arange(n, dtype=torch.int64, device=state.device)
        else:
            raise ValueError(f"action_code batch does not match state batch size: state_batch={state.batch_size}, "
                             f"action_batch={n}")
        return batch_apply_moves_compat(state, codes, parents)

    @staticmethod
    def _normalize_temperatures(temperatures, batch_size: int, default_temperature: float, device) -> torch.Tensor:
        if temperatures is None:
            return torch.full((batch_size,), float(default_temperature), dtype=torch.float32, device=device)
        if isinstance(temperatures, (float, int)):
            return torch.full((batch_size,), float(temperatures), dtype=torch.float32, device=device)
        t = torch.as_tensor(temperatures, dtype=torch.float32, device=device).view(-1)
        if int(t.numel()) != batch_size:
            raise ValueError(f"temperatures size mismatch: expected {batch_size}, got {int(t.numel())}")
        return t

    # ---- children of a batch of positions, valued from the parent mover's side ----
    def _children_leaf_matrix(self, state: GpuStateBatch, codes_all, parents_all, flat_idx, R: int, M: int) -> torch.Tensor:
        """leaf_mat [R, M]: value of every packed child as its parent's mover sees it; a child that ends the game
        takes the soft piece-count value instead of the network's (mcts_gpu.py:1341-1375)."""
        cfg = self.config
        child = batch_apply_moves_compat(state, codes_all, parents_all)
        if self._child_eval_mode == "full":
            child_values = self._evaluate_batch(child)[4]
        else:
            child_values = self._evaluate_values_only(child)
        parent_player = state.current_player.index_select(0, parents_all)
        leaf = self._child_values_to_parent_perspective(child_values, parent_player, child.current_player)
        term_child = self._terminal_mask_from_next_state(child)        # sync-free here
        soft_black = self._soft_tanh_from_board_black(child.board, float(cfg.soft_value_k))
        sign = torch.where(parent_player.ge(0), 1.0, -1.0).to(torch.float32)
        leaf = torch.where(term_child, soft_black * sign, leaf)
        leaf_mat = torch.zeros((R, M), dtype=torch.float32, device=state.device)
        leaf_mat.view(-1).index_copy_(0, flat_idx, leaf)
        return leaf_mat

    def _refine_via_topk_lookahead(self, root_states: GpuStateBatch, leaf_mat: torch.Tensor, valid_mask: torch.Tensor,
                                   code_mat: torch.Tensor) -> torch.Tensor:
        """One ply deeper below the K best children of every root (mcts_gpu.py:976-1046): a child's value becomes
        max(its value, the best value among its own children as its mover sees them).  Slots beyond a root's legal
        count carry no action: they re-evaluate the root's first action and their result is dropped."""
        R, M = int(valid_mask.shape[0]), int(valid_mask.shape[1])
        K = min(int(self.config.sparse_top_k), M)
        if K <= 0 or R <= 0:
            return leaf_mat
        dev = leaf_mat.device
        top = torch.topk(leaf_mat.masked_fill(~valid_mask, float("-inf")), k=K, dim=1).indices      # [R, K]
        picked = valid_mask.gather(1, top)
        safe = torch.where(picked, top, top[:, :1].expand(-1, K))
        l2_codes = code_mat.gather(1, safe.unsqueeze(-1).expand(-1, -1, 4)).reshape(-1, 4).contiguous()
        parents = torch.arange(R, dtype=torch.int64, device=dev).repeat_interleave(K)
        l2 = batch_apply_moves_compat(root_states, l2_codes, parents)
        _, l2_mask, l2_meta, l2_probs, _ = self._evaluate_batch(l2)
        (_, l2_roots, _, l2_valid, _, _, _, l2_flat, l2_codes_all, l2_parents_all) = \
            v0_core.root_pack_sparse_actions(l2_mask, l2_probs, l2_meta)
        refined = torch.zeros((R * K,), dtype=torch.float32, device=dev)          # no grandchildren -> 0
        if int(l2_roots.numel()) > 0:
            l3 = self._children_leaf_matrix(l2, l2_codes_all, l2_parents_all, l2_flat, int(l2_valid.shape[0]),
                                            int(l2_valid.shape[1]))
            best = l3.masked_fill(~l2_valid, float("-inf")).max(dim=1).values
            refined.index_copy_(0, l2_roots, torch.where(torch.isfinite(best), best, torch.zeros_like(best)))
        new = torch.maximum(leaf_mat.gather(1, top), refined.view(R, K))
        keep = leaf_mat.gather(1, top)
        return leaf_mat.scatter(1, top, torch.where(picked, new, keep))

    # ---- the search ----
    def search_batch(self, state: GpuStateBatch, *, temperatures=None, add_dirichlet_noise: Optional[bool] = None,
                     force_uniform_random_mask: Optional[torch.Tensor] = None) -> RootSearchBatchOutput:
        cfg = self.config
        dev = state.device
        B = int(state.batch_size)
        add_noise = cfg.add_dirichlet_noise if add_dirichlet_noise is None else bool(add_dirichlet_noise)
        force_mask = None
        if force_uniform_random_mask is not None:
            force_mask = torch.as_tensor(force_uniform_random_mask, device=dev).to(torch.bool).view(-1)
            if int(force_mask.numel()) != B:
                raise ValueError(f"force_uniform_random_mask size mismatch: expected {B}, got {int(force_mask.numel())}")
        temps = self._normalize_temperatures(temperatures, B, cfg.temperature, dev)

        model_input, legal_mask, metadata, probs, values = self._evaluate_batch(state)
        root_values = values.clone()

        with self._timed("pack_writeback_ms"):
            (terminal_mask, roots, counts, valid_mask, lidx_mat, priors_mat, code_mat, flat_idx, codes_all,
             parents_all) = v0_core.root_pack_sparse_actions(legal_mask, probs, metadata)

        policy_dense = torch.zeros((B, TOTAL_ACTION_DIM), dtype=torch.float32, device=dev)
        chosen_idx = torch.full((B,), -1, dtype=torch.int64, device=dev)
        chosen_codes = torch.full((B, 4), -1, dtype=torch.int32, device=dev)
        chosen_valid = torch.zeros((B,), dtype=torch.bool, device=dev)

        R = int(roots.numel())
        if R > 0:
            M = int(valid_mask.shape[1])
            if add_noise and M > 1:                                     # mcts_gpu.py:1329-1339
                if self.injected_noise is not None:
                    noise = self.injected_noise.to(dev, torch.float32)
                else:
                    alpha = torch.full_like(priors_mat, float(cfg.dirichlet_alpha))
                    noise = torch._standard_gamma(alpha)
                noise = noise * valid_mask.to(torch.float32)
                noise = noise / noise.sum(dim=1, keepdim=True).clamp_min(1e-8)
                eps = float(cfg.dirichlet_epsilon)
                mixed = (1.0 - eps) * priors_mat + eps * noise
                priors_mat = torch.where(counts.gt(1).view(-1, 1), mixed, priors_mat)

            leaf_mat = self._children_leaf_matrix(state, codes_all, parents_all, flat_idx, R, M)
            if int(cfg.sparse_ply) > 1:                                 # experimental top-K lookahead, :1150-1160
                root_states = state.select(roots)
                for _ in range(2, int(cfg.sparse_ply) + 1):
                    leaf_mat = self._refine_via_topk_lookahead(root_states, leaf_mat, valid_mask, code_mat)

            sims = max(1, int(cfg.num_simulations))
            with self._timed("root_puct_ms"):
                visits, value_sum, _ = v0_core.root_puct_allocate_visits(priors_mat, leaf_mat, valid_mask, sims,
                                                                         float(cfg.exploration_weight))
            root_temps = temps.index_select(0, roots)
            sample = bool(cfg.sample_moves and M > 1)
            with self._timed("pack_writeback_ms"):
                policy_dense, chosen_idx, chosen_codes, chosen_valid, root_value_vec = \
                    v0_core.root_finalize_from_visits(lidx_mat, code_mat, valid_mask, visits, value_sum, roots, B,
                                                      TOTAL_ACTION_DIM, root_temps, sample,
                                                      uniforms=self.injected_uniforms)
            if force_mask is not None:                                  # mcts_gpu.py:1425-1445
                f_local = force_mask.index_select(0, roots)
                f_rows = torch.nonzero(f_local).view(-1)
                if int(f_rows.numel()) > 0:
                    vm = valid_mask.index_select(0, f_rows).to(torch.float32)
                    picks = torch.multinomial(vm / vm.sum(dim=1, keepdim=True).clamp_min(1e-8), 1).view(-1)
                    g_roots = roots.index_select(0, f_rows)
                    chosen_idx.index_copy_(0, g_roots, lidx_mat.index_select(0, f_rows).gather(1, picks.view(-1, 1)).view(-1))
                    chosen_codes.index_copy_(0, g_roots, code_mat.index_select(0, f_rows).gather(
                        1, picks.view(-1, 1, 1).expand(-1, 1, 4)).view(-1, 4))
                    chosen_valid.index_fill_(0, g_roots, True)
                    self._forced_uniform_pick_count += int(f_rows.numel())
            root_values.index_copy_(0, roots, root_value_vec)

        return RootSearchBatchOutput(
            model_input=model_input, legal_mask=legal_mask, policy_dense=policy_dense, root_value=root_values,
            terminal_mask=terminal_mask, chosen_action_indices=chosen_idx, chosen_action_codes=chosen_codes,
            chosen_valid_mask=chosen_valid)

    def search(self, state: GpuStateBatch, *, temperature: Optional[float] = None,
               add_dirichlet_noise: Optional[bool] = None) -> RootSearchOutput:
        if state.batch_size != 1:
            raise ValueError("V1RootMCTS.search currently supports a single root state.")
        out = self.search_batch(state, temperatures=self.config.temperature if temperature is None else float(temperature),
                                add_dirichlet_noise=add_dirichlet_noise)
        ok = bool(out.chosen_valid_mask[0].item())
        return RootSearchOutput(
            model_input=out.model_input[0], legal_mask=out.legal_mask[0], policy_dense=out.policy_dense[0],
            root_value=float(out.root_value[0].item()), terminal=bool(out.terminal_mask[0].item()),
            chosen_action_index=int(out.chosen_action_indices[0].item()) if ok else None,
            chosen_action_code=out.chosen_action_codes[0] if ok else None)
