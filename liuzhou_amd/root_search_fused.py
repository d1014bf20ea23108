"""Fused root-PUCT search (variant R): the whole of `V1RootMCTS.search_batch` as a fixed-shape, sync-free sequence of
launches on packed states -- pack, network (roots), `lz_root_prepare`, network (children, batch size read on the
device, values only), `lz_root_collect`, `lz_root_puct_allocate_visits`, `lz_root_finalize_from_visits` -- captured in
one hipGraph.  Same outputs as `V1RootMCTS.search_batch` (asserted against it in tests/test_gpu_selfplay.py)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L
from .game_rng import GameRng, PURPOSE_OPENING, PURPOSE_PICK
from .mcts_gpu import GpuStateBatch, RootSearchBatchOutput, TOTAL_ACTION_DIM, encode_actions_fast, states_to_model_input
from .net_hip import FusedNet
from .streams import CAPTURE_MODE

CAP = 72


class FusedRootSearch:
    def __init__(self, net: FusedNet, num_games: int, num_simulations: int, device, exploration_weight: float = 1.0,
                 add_dirichlet_noise: bool = True, dirichlet_alpha: float = 0.3, dirichlet_epsilon: float = 0.25,
                 sample_moves: bool = True, soft_value_k: float = 2.0, use_graph: bool = True, out=None,
                 seed: int = 12345, game_offset: int = 0, game_stride: Optional[int] = None,
                 sparse_ply: int = 1, sparse_top_k: int = 8, child_eval_mode: str = "value_only") -> None:
        """`out`: optional dict of preallocated output tensors (rows of a larger batch: DualStreamRootSearch).
        `sparse_ply` > 1 / `sparse_top_k`: the reference's top-K lookahead (v1/python/mcts_gpu.py:976-1046, :1150-1160) as
        `sparse_ply - 1` extra fixed-shape rounds over B x top_k L2 positions -- lz_root_topk_children, network (full),
        lz_root_prepare, network (children, counted), lz_root_collect, lz_root_refine_topk -- inside the same captured
        launch sequence (round 6; before, these options dropped to the host-synced operator chain).
        `child_eval_mode` "full" (:1342-1345): the children go through the whole network instead of the value head only;
        the values are the same bits (tests/test_gpu_net.py::test_values_only_mode_equals_full_forward), the policy rows
        are computed and dropped as the reference does.
        `seed` / `game_offset` / `game_stride`: keys of the per-game counter RNG (game_rng.GameRng) that replaces the
        reference's draws from the device generator (v1/python/mcts_gpu.py:1329-1339,1410-1424)."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("FusedRootSearch needs a HIP device (no CPU path)")
        self.net, self.B, self.sims, self.device = net, int(num_games), max(1, int(num_simulations)), dev
        self.sparse_ply, self.top_k = max(1, int(sparse_ply)), max(1, min(int(sparse_top_k), CAP))
        self.child_eval_mode = str(child_eval_mode).strip().lower()
        if self.child_eval_mode not in ("value_only", "full"):
            raise ValueError(f"Unsupported child_eval_mode={child_eval_mode!r}; expected one of ('value_only', 'full').")
        self.c, self.add_noise = float(exploration_weight), bool(add_dirichlet_noise)
        self.alpha, self.eps, self.sample_moves, self.soft_k = float(dirichlet_alpha), float(dirichlet_epsilon), bool(sample_moves), float(soft_value_k)
        self.use_graph = bool(use_graph) and os.environ.get("LZ_ROOT_GRAPH", "on").strip().lower() not in ("off", "0", "false")
        B = self.B
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        out = out or {}
        o = lambda name, shape, dt: out[name] if name in out else z(shape, dt)
        self.root_packed = z((B, 4), torch.int64)
        self.lp1, self.lp2, self.lpm = (z((B, 36), torch.float32) for _ in range(3))
        self.values = o("values", (B,), torch.float32)
        self.legal_index = z((B, CAP), torch.int64)
        self.priors = z((B, CAP), torch.float32)
        self.codes = z((B, CAP, 4), torch.int32)
        self.valid = z((B, CAP), torch.uint8)
        self.counts = o("counts", (B,), torch.int32)
        self.terminal = o("terminal", (B,), torch.uint8)
        self.leaf = z((B, CAP), torch.float32)
        self.child_states = z((B * CAP, 4), torch.int64)
        self.child_ref = z((B * CAP,), torch.int32)
        self.child_values = z((B * CAP,), torch.float32)
        self.n_children = z((1,), torch.int64)
        self.overflow = z((1,), torch.int32)
        self.visits, self.value_sum = z((B, CAP), torch.float32), z((B, CAP), torch.float32)
        self.puct_root_values = z((B,), torch.float32)
        self.roots = torch.arange(B, dtype=torch.int64, device=dev)
        self.temps = z((B,), torch.float32)
        self.noise = z((B, CAP), torch.float32)
        self.uniforms = z((B,), torch.float32)
        self.force = z((B,), torch.uint8)                   # opening plies: the move is a uniform pick (search_batch)
        self.force_uniforms = z((B,), torch.float32)
        self.rng = GameRng(B, dev, seed=seed, game_offset=game_offset, game_stride=game_stride)
        self.policy_dense = o("policy_dense", (B, TOTAL_ACTION_DIM), torch.float32)
        self.chosen_idx = o("chosen_idx", (B,), torch.int64)
        self.chosen_codes = o("chosen_codes", (B, 4), torch.int32)
        self.chosen_valid = o("chosen_valid", (B,), torch.uint8)
        self.root_value_vec = o("root_value_vec", (B,), torch.float32)
        self._graphs = {}
        self._evals_dev = z((1,), torch.int64)              # network evaluations so far, kept on the device
        full = self.child_eval_mode == "full"
        # policy rows of the children (child_eval_mode "full": computed, never read)
        self.child_lp = tuple(z((B * CAP, 36), torch.float32) for _ in range(3)) if full else None
        if self.sparse_ply > 1:
            K = self.top_k
            N2 = B * K                                                       # L2 positions per lookahead round
            self.top_slot = z((B, K), torch.int32)
            self.l2_states = z((N2, 4), torch.int64)
            self.l2_lp = tuple(z((N2, 36), torch.float32) for _ in range(3))
            self.l2_values = z((N2,), torch.float32)
            self.l2_legal_index = z((N2, CAP), torch.int64)
            self.l2_priors = z((N2, CAP), torch.float32)
            self.l2_codes = z((N2, CAP, 4), torch.int32)
            self.l2_valid = z((N2, CAP), torch.uint8)
            self.l2_counts, self.l2_terminal = z((N2,), torch.int32), z((N2,), torch.uint8)
            self.l2_leaf = z((N2, CAP), torch.float32)
            self.l3_states = z((N2 * CAP, 4), torch.int64)
            self.l3_ref = z((N2 * CAP,), torch.int32)
            self.l3_values = z((N2 * CAP,), torch.float32)
            self.n_l3 = z((1,), torch.int64)
            self.l3_total = z((1,), torch.int64)                             # grandchildren evaluated by the last search
            self.l3_lp = tuple(z((N2 * CAP, 36), torch.float32) for _ in range(3)) if full else None
        # lists of the width-binned bandit (rows of <= 16 / <= 32 actions share a wave four / two at a time): caller-owned,
        # so that the launch is allocation-free and can be captured on whatever stream torch captures on
        ws_bytes = C.c_int64(0)
        L.check(L.lib().lz_root_puct_workspace_bytes(L.i64(B), C.byref(ws_bytes)), "root_puct_workspace_bytes")
        self.puct_ws = z((int(ws_bytes.value),), torch.uint8)

    def _launch(self, add_noise: bool, sample: bool, forced: bool = False) -> None:
        lib, st, B, p = L.lib(), L.stream_ptr(self.device), self.B, L.ptr
        ck = L.check
        ck(lib.lz_net_forward_packed_f16(C.byref(self.net.desc), p(self.root_packed), L.i64(B), p(self.lp1), p(self.lp2),
                                         p(self.lpm), None, p(self.values), st), "net_forward_packed")
        ck(lib.lz_root_prepare(p(self.root_packed), L.i64(B), p(self.lp1), p(self.lp2), p(self.lpm),
                               p(self.noise) if add_noise else None, C.c_float(self.eps), p(self.legal_index), p(self.priors),
                               p(self.codes), p(self.valid), p(self.counts), p(self.terminal), p(self.leaf),
                               p(self.child_states), p(self.child_ref), p(self.n_children), L.i64(B * CAP),
                               p(self.overflow), st), "root_prepare")
        clp = [p(t) for t in self.child_lp] if self.child_lp is not None else [None, None, None]
        ck(lib.lz_net_forward_packed_counted_f16(C.byref(self.net.desc), p(self.child_states), L.i64(B * CAP),
                                                 p(self.n_children), clp[0], clp[1], clp[2], None, p(self.child_values), st),
           "net_forward_packed_counted")
        ck(lib.lz_root_collect(p(self.root_packed), p(self.child_states), p(self.child_ref), p(self.child_values),
                               p(self.n_children), L.i64(B * CAP), C.c_float(self.soft_k), p(self.leaf), st), "root_collect")
        if self.sparse_ply > 1:
            self.l3_total.mul_(0)
        for _ply in range(2, self.sparse_ply + 1):          # top-K lookahead (mcts_gpu.py:1150-1160): fixed shapes, no sync
            K, N2 = self.top_k, B * self.top_k
            ck(lib.lz_root_topk_children(p(self.root_packed), L.i64(B), p(self.leaf), p(self.valid), p(self.codes), L.i64(K),
                                         p(self.top_slot), p(self.l2_states), st), "root_topk_children")
            ck(lib.lz_net_forward_packed_f16(C.byref(self.net.desc), p(self.l2_states), L.i64(N2), p(self.l2_lp[0]),
                                             p(self.l2_lp[1]), p(self.l2_lp[2]), None, p(self.l2_values), st),
               "net_forward_packed(L2)")
            ck(lib.lz_root_prepare(p(self.l2_states), L.i64(N2), p(self.l2_lp[0]), p(self.l2_lp[1]), p(self.l2_lp[2]), None,
                                   C.c_float(0.0), p(self.l2_legal_index), p(self.l2_priors), p(self.l2_codes),
                                   p(self.l2_valid), p(self.l2_counts), p(self.l2_terminal), p(self.l2_leaf),
                                   p(self.l3_states), p(self.l3_ref), p(self.n_l3), L.i64(N2 * CAP), p(self.overflow), st),
               "root_prepare(L2)")
            l3lp = [p(t) for t in self.l3_lp] if self.l3_lp is not None else [None, None, None]
            ck(lib.lz_net_forward_packed_counted_f16(C.byref(self.net.desc), p(self.l3_states), L.i64(N2 * CAP), p(self.n_l3),
                                                     l3lp[0], l3lp[1], l3lp[2], None, p(self.l3_values), st),
               "net_forward_packed_counted(L3)")
            ck(lib.lz_root_collect(p(self.l2_states), p(self.l3_states), p(self.l3_ref), p(self.l3_values), p(self.n_l3),
                                   L.i64(N2 * CAP), C.c_float(self.soft_k), p(self.l2_leaf), st), "root_collect(L2)")
            ck(lib.lz_root_refine_topk(L.i64(B), L.i64(K), p(self.top_slot), p(self.l2_leaf), p(self.l2_valid), p(self.leaf),
                                       st), "root_refine_topk")
            self.l3_total.add_(self.n_l3)                    # (an elementwise kernel node: no memset / memcpy nodes in the graph)
        ck(lib.lz_root_puct_allocate_visits_ws(p(self.priors), p(self.leaf), p(self.valid), L.i64(B), L.i64(CAP),
                                               L.i64(self.sims), C.c_float(self.c), p(self.visits), p(self.value_sum),
                                               p(self.puct_root_values), p(self.puct_ws), L.i64(int(self.puct_ws.numel())),
                                               st), "root_puct")
        ck(lib.lz_root_finalize_from_visits(p(self.legal_index), p(self.codes), p(self.valid), p(self.visits),
                                            p(self.value_sum), p(self.roots), L.i64(B), L.i64(CAP), L.i64(B),
                                            L.i64(TOTAL_ACTION_DIM), p(self.temps), p(self.uniforms) if sample else None,
                                            p(self.policy_dense), p(self.chosen_idx), p(self.chosen_codes),
                                            p(self.chosen_valid), p(self.root_value_vec), st), "root_finalize")
        if forced:                                           # opening plies (mcts_gpu.py:1425-1447): uniform picks replace the search's
            ck(lib.lz_root_force_uniform_picks(p(self.legal_index), p(self.codes), p(self.valid), p(self.roots), L.i64(B),
                                               L.i64(CAP), p(self.force), p(self.force_uniforms), p(self.chosen_idx),
                                               p(self.chosen_codes), p(self.chosen_valid), st), "root_force_uniform_picks")

    def search_batch(self, state: GpuStateBatch, *, temperatures: torch.Tensor, add_dirichlet_noise: Optional[bool] = None,
                     injected_noise: Optional[torch.Tensor] = None, injected_uniforms: Optional[torch.Tensor] = None,
                     want_output: bool = True, reset: Optional[torch.Tensor] = None,
                     rng_game_ids: Optional[torch.Tensor] = None,
                     rng_plies: Optional[torch.Tensor] = None,
                     force_uniform_random_mask: Optional[torch.Tensor] = None,
                     injected_force_uniforms: Optional[torch.Tensor] = None) -> Optional[RootSearchBatchOutput]:
        """`reset` uint8[B]: slots whose game was re-seated since the last search (their RNG key moves on to the next
        game id); `rng_game_ids` / `rng_plies`: the runner's own numbering instead.
        `force_uniform_random_mask` bool[B] (the reference's opening plies, v1/python/mcts_gpu.py:1255-1272,1425-1447):
        flagged games play a uniformly random legal move, drawn from the per-game RNG (purpose OPENING); their policy
        target is still the search's."""
        B, dev = self.B, self.device
        if int(state.batch_size) != B:
            raise ValueError(f"FusedRootSearch was built for {B} games, got {int(state.batch_size)}")
        add_noise = self.add_noise if add_dirichlet_noise is None else bool(add_dirichlet_noise)
        ts = [t if t.is_contiguous() else t.contiguous() for t in state.tensors()]
        with torch.cuda.device(dev):
            L.check(L.lib().lz_pack_states(C.byref(L.soa(ts)), L.i64(B), L.ptr(self.root_packed), L.stream_ptr(dev)), "pack_states")
        self.temps.copy_(temperatures.to(torch.float32).reshape(-1))
        self.rng.begin_move(reset, rng_game_ids, rng_plies)
        if add_noise:
            if injected_noise is not None:
                self.noise.zero_()
                self.noise[:, : injected_noise.shape[1]].copy_(injected_noise.to(torch.float32))
            else:
                self.rng.gamma_into(self.noise, self.alpha, CAP)
        sample = self.sample_moves
        if sample:
            if injected_uniforms is not None:
                self.uniforms.copy_(injected_uniforms.to(torch.float32))
            else:
                self.rng.uniform_into(self.uniforms, PURPOSE_PICK)
        forced = force_uniform_random_mask is not None
        if forced:
            self.force.copy_(force_uniform_random_mask.reshape(-1).to(torch.uint8))
            if injected_force_uniforms is not None:
                self.force_uniforms.copy_(injected_force_uniforms.to(torch.float32))
            else:
                self.rng.uniform_into(self.force_uniforms, PURPOSE_OPENING)
        self.rng.end_move(rng_plies is not None)
        key = (add_noise, sample, forced)
        with torch.cuda.device(dev):
            if not self.use_graph:
                self._launch(add_noise, sample, forced)
            else:
                g = self._graphs.get(key)
                if g is None:
                    self._launch(add_noise, sample, forced)          # warm-up (idempotent: every buffer is rewritten)
                    torch.cuda.synchronize(dev)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                        self._launch(add_noise, sample, forced)
                    self._graphs[key] = g
                g.replay()
        self._evals_dev.add_(self.n_children).add_(B)
        if self.sparse_ply > 1:
            self._evals_dev.add_(self.l3_total).add_((self.sparse_ply - 1) * B * self.top_k)
        if not want_output:
            return None
        has_root = self.counts > 0
        root_values = torch.where(has_root, self.root_value_vec, self.values)
        model_input = states_to_model_input(state)
        legal_mask, _ = encode_actions_fast(state)
        return RootSearchBatchOutput(
            model_input=model_input, legal_mask=legal_mask, policy_dense=self.policy_dense, root_value=root_values,
            terminal_mask=self.terminal.view(torch.bool), chosen_action_indices=self.chosen_idx,
            chosen_action_codes=self.chosen_codes, chosen_valid_mask=self.chosen_valid.view(torch.bool))

    @property
    def leaf_evals(self) -> int:
        """Root + child evaluations of all searches so far (one host read)."""
        return int(self._evals_dev.item())

    def children_evaluated(self) -> int:
        """Number of child evaluations of the last search (host read: not for the hot loop)."""
        return int(self.n_children.item())



class DualStreamRootSearch:
    """Two FusedRootSearch halves on two HIP streams, same interface.  The kernels around the network launches (legal
    sets and priors, the bandit, the policy extraction: ~30 % of a search at C2) are latency-bound and leave the
    matrix pipes idle; with the games split in two halves they overlap the other half's network launches.  The halves
    write rows of shared output tensors, so the result is the one FusedRootSearch gives over all games."""

    def __init__(self, net: FusedNet, num_games: int, num_simulations: int, device, **kw) -> None:
        dev = torch.device(device)
        self.B, self.device = int(num_games), dev
        B = self.B
        h = (B + 1) // 2
        self.bounds = ((0, h), (h, B))
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.out = {"values": z((B,), torch.float32), "counts": z((B,), torch.int32), "terminal": z((B,), torch.uint8),
                    "policy_dense": z((B, TOTAL_ACTION_DIM), torch.float32), "chosen_idx": z((B,), torch.int64),
                    "chosen_codes": z((B, 4), torch.int32), "chosen_valid": z((B,), torch.uint8),
                    "root_value_vec": z((B,), torch.float32)}
        kw.pop("game_offset", None); kw.pop("game_stride", None)
        self.parts = [FusedRootSearch(net, b - a, num_simulations, dev, out={k: v[a:b] for k, v in self.out.items()},
                                      game_offset=a, game_stride=B, **kw)
                      for a, b in self.bounds]
        from .streams import overlapping_streams
        self.streams = overlapping_streams(dev, len(self.parts))
        self.serialize = False        # measurement aid: run the halves one after the other on the caller's stream

    @property
    def use_graph(self) -> bool:
        return self.parts[0].use_graph

    @use_graph.setter
    def use_graph(self, v: bool) -> None:
        for p in self.parts:
            p.use_graph = bool(v)

    @property
    def leaf_evals(self) -> int:
        return sum(p.leaf_evals for p in self.parts)

    @property
    def overflow(self) -> torch.Tensor:
        return self.parts[0].overflow + self.parts[1].overflow

    def search_batch(self, state: GpuStateBatch, *, temperatures: torch.Tensor, add_dirichlet_noise: Optional[bool] = None,
                     injected_noise: Optional[torch.Tensor] = None, injected_uniforms: Optional[torch.Tensor] = None,
                     reset: Optional[torch.Tensor] = None, rng_game_ids: Optional[torch.Tensor] = None,
                     rng_plies: Optional[torch.Tensor] = None,
                     force_uniform_random_mask: Optional[torch.Tensor] = None) -> RootSearchBatchOutput:
        if int(state.batch_size) != self.B:
            raise ValueError(f"DualStreamRootSearch was built for {self.B} games, got {int(state.batch_size)}")
        main = torch.cuda.current_stream(self.device)
        cut = lambda t, a, b: None if t is None else t[a:b]
        for (a, b), part, st in zip(self.bounds, self.parts, (main, main) if self.serialize else self.streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                part.search_batch(state._map(lambda t, a=a, b=b: t[a:b]), temperatures=temperatures[a:b],
                                  add_dirichlet_noise=add_dirichlet_noise, injected_noise=cut(injected_noise, a, b),
                                  injected_uniforms=cut(injected_uniforms, a, b), want_output=False,
                                  reset=cut(reset, a, b), rng_game_ids=cut(rng_game_ids, a, b),
                                  rng_plies=cut(rng_plies, a, b),
                                  force_uniform_random_mask=cut(force_uniform_random_mask, a, b))
        if not self.serialize:
            for st in self.streams:
                main.wait_stream(st)
        o = self.out
        root_values = torch.where(o["counts"] > 0, o["root_value_vec"], o["values"])
        model_input = states_to_model_input(state)
        legal_mask, _ = encode_actions_fast(state)
        return RootSearchBatchOutput(
            model_input=model_input, legal_mask=legal_mask, policy_dense=o["policy_dense"], root_value=root_values,
            terminal_mask=o["terminal"].view(torch.bool), chosen_action_indices=o["chosen_idx"],
            chosen_action_codes=o["chosen_codes"], chosen_valid_mask=o["chosen_valid"].view(torch.bool))
