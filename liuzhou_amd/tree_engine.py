"""Device-resident full-tree PUCT search (variant P) -- host side.

`TreeEngine` owns the HBM arenas of B concurrent search trees (csrc/lz_engine.hip) and exposes the same
split-phase protocol as the reference's C++ tree batch (`v1/python/portable_cpp_mcts.py:270-282`:
prepare roots -> evaluate -> complete, then sims x (select -> evaluate -> complete)), plus
`search()` which enqueues a whole move's search from C++ with the fused network kernel in the loop.

`PortableTreeMCTS.search_batch(state, ...)` gives it the `V1RootMCTS`-style interface (same
`RootSearchBatchOutput`), `self_play_tree_gpu` is the tree-search twin of `self_play_v1_gpu`, and
`SteadyStateTreeSelfPlay` is the fixed-population driver used by bench.py.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time
from typing import Dict, List, Optional, Tuple

import torch

from contextlib import contextmanager

from . import _lib as L
from . import v0_core
from .game_rng import GameRng, PURPOSE_PICK
from .mcts_gpu import GpuStateBatch, RootSearchBatchOutput, TOTAL_ACTION_DIM, states_to_model_input, \
    encode_actions_fast
from .net_hip import FusedNet, LzNetDesc
from .self_play_types import SelfPlayV1Stats
from .streams import CAPTURE_MODE
from .trajectory_buffer import TensorSelfPlayBatch, TensorTrajectoryBuffer

MAX_CHILDREN = 72
OUT_CAP = 80


class LzTreeDesc(C.Structure):
    _fields_ = [("num_games", C.c_int64), ("node_cap", C.c_int32), ("edge_chunk", C.c_int32), ("path_cap", C.c_int32),
                ("chunk_cap", C.c_int32), ("exploration_weight", C.c_double)] + \
               [(n, C.c_void_p) for n in (
                   "root_state", "nodes", "edges", "n_nodes", "n_edges", "root_visits", "root_w",
                   "root_init_value", "path", "path_len", "leaf_kind", "leaf_state", "leaf_value", "root_terminal",
                   "active", "leaf_edge", "leaf_parent",
                   "trace_kind", "trace_leaf", "trace_heads", "trace_priors", "trace_value")] + \
               [("trace_cap", C.c_int64), ("eval_count", C.c_void_p)] + \
               [(n, C.c_void_p) for n in ("chunk_list", "n_chunks", "free_chunks", "pool_top", "pool_stats")] + \
               [("pool_chunks", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("live_state", "live_row", "live_count")] + [("live_count_cap", C.c_int64)]


class LzTreeWaveDesc(C.Structure):
    _fields_ = [("batch_k", C.c_int32), ("path_cap", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("path", "path_len", "leaf_kind", "leaf_state", "leaf_value", "leaf_edge",
                                          "leaf_parent", "sims_done", "unfinished", "eval_row", "eval_state",
                                          "eval_count", "eval_total")] + \
               [("max_backtrack_steps", C.c_int32), ("reserved_", C.c_int32)]


WAVE_PATH_CAP = 176            # entries per leaf path of a wave (the kernel's level stack holds 160: a game lasts <= 144 plies)
EDGE_CHUNK = 1024              # edges per chunk of the engine's edge pool (32 KB); a run of <= 72 edges never straddles chunks
# pool sizing: 3 x sims nodes of 32 edges per game.  Measured peak use (bench.py `reuse.edge_pool.peak_use_frac`, round 4):
# 15 edges per simulation of the budget at C3 (16 384 x 800) and 17 at C2 -- a steady-state population is mostly in the
# movement phase with 10-20 legal moves, and a kept subtree rarely exceeds the new search -- so 96 is 6 x head-room
POOL_EDGES_PER_NODE = 32
POOL_NODES_PER_SIM = 3
MAX_NODE_CAP = 524288          # lz_tree_advance marks a game's nodes in LDS: 8 192 words of 64 with one wave per workgroup
                               # (round 6; four waves per workgroup up to 65 536 nodes, the limit until round 5)
AUTO_NODE_CAP = 65536          # arenas sized automatically (auto_reuse_factor) stay within the four-wave form
PATH_CAP = 192                 # a game lasts <= 144 plies (game_state.py:87-89), so no descent is deeper than that
REUSE_FACTOR_CAP = 40.0


def auto_reuse_factor(num_games: int, sims: int, device, memory_fraction: float = 0.10, cap: float = REUSE_FACTOR_CAP) -> float:
    """Room for kept subtrees in the per-game NODE arena, as a multiple of `sims` nodes: a kept subtree that would leave
    no room for the next search is pruned, so more room = fewer deviations from the reference's unbounded tree (runs of
    near-forced moves keep almost the whole tree several times in a row).  Since round 4 only the 48-byte node records
    are per game (edges come from the engine's pool), so the factor is `cap` unless `memory_fraction` of the free memory
    or the 65 536-node limit say less (C3: 40 -> 32 802 nodes = 1.5 MB per game, 25.8 GB; rounds 1-3 had worst-case edge
    regions per game and reached 11-12 with 210 GB)."""
    free, _total = torch.cuda.mem_get_info(torch.device(device))
    per_game = free * float(memory_fraction) / max(1, int(num_games))
    f = (per_game / 48.0 - (sims + 2)) / max(1, sims)
    f = min(float(cap), f, (AUTO_NODE_CAP - sims - 2) / max(1, sims))
    return max(1.0, float(int(f * 4) / 4.0))


def chunk_cap_for(node_cap: int, chunk: int) -> int:
    """Entries of a game's chunk list: the worst case of its node arena (72 children everywhere), so that only the node
    arena bounds a single game (include/liuzhou_hip.h)."""
    return -(-int(node_cap) * MAX_CHILDREN // (int(chunk) - (MAX_CHILDREN - 1))) + 1


def auto_pool_chunks(num_games: int, max_sims: int, node_cap: int, chunk: int, device, memory_fraction: float = 0.5) -> int:
    """Chunks of the engine's edge pool.  Never more than the worst case (every node of every game with 72 children:
    then an allocation cannot fail, which is what small engines get); large engines get the MEAN case with head-room --
    POOL_NODES_PER_SIM x sims nodes x POOL_EDGES_PER_NODE edges per game, at least 64 worst-case games -- because the sum
    over thousands of games is stable although single games vary by 20x; bounded by `memory_fraction` of the free memory
    and by the 2^31 pool indices (64 GB)."""
    worst = int(num_games) * chunk_cap_for(node_cap, chunk)
    # per game: the mean case rounded up to whole chunks, + 1 for the open, partly filled chunk every game holds
    mean = int(num_games) * (-(-min(int(node_cap), POOL_NODES_PER_SIM * (int(max_sims) + 1)) * POOL_EDGES_PER_NODE // int(chunk)) + 1)
    floor = min(int(num_games), 64) * chunk_cap_for(node_cap, chunk)
    n = min(worst, max(mean, floor))
    free, _total = torch.cuda.mem_get_info(torch.device(device))
    n = min(n, int(free * float(memory_fraction)) // (int(chunk) * 32), (1 << 31) // int(chunk) - 2)
    return max(n, 2)


def samples_per_launch_pass(net) -> int:
    """Positions one pass of the fused network kernel evaluates over the whole chip: workgroups per launch x samples per
    workgroup (csrc/lz_net.hip: 256 x 16 for 64 channels, 512 x 8 in the half-workgroup shape, 256 x 8 for 128 channels)."""
    ch, flags = int(net.desc.channels), int(net.desc.flags)
    groups = int(net.desc.max_blocks) if int(net.desc.max_blocks) > 0 else (512 if (ch == 64 and flags & 1) else 256)
    return groups * (16 if (ch == 64 and not flags & 1) else 8)


class TreeEngine:
    def __init__(self, num_games: int, max_sims: int, device, exploration_weight: float = 1.0,
                 reuse_factor: float = 0.0, batch_k: int = 1, edge_chunk: int = EDGE_CHUNK,
                 pool_chunks: Optional[int] = None, compact_evals: bool = False, max_backtrack_steps: int = 0) -> None:
        """`max_backtrack_steps` (batch_k > 1): the reference's MAX_BACKTRACK_STEPS (src/mcts.py:337; 0 = its 128).
        `compact_evals`: the fused search evaluates, per simulation, only the leaves that need the network (compact
        device-side list, `LzTreeDesc.live_*`) -- a launch then costs ceil(live / samples per pass) network passes instead
        of one per slot, so a draining wave gets cheaper as its games end; bit-identical results.
        `reuse_factor` > 0 reserves room for kept subtrees of up to reuse_factor * max_sims nodes in every game's node
        arena (advance(); < 0: auto_reuse_factor).  Edges live in ONE pool per engine, handed out in chunks of `edge_chunk`
        records (`pool_chunks`: None = auto_pool_chunks).
        `batch_k` > 1: the legacy search's waves (src/mcts.py `batch_K`): up to batch_k distinct leaves per game are
        collected, evaluated together and backed up per wave (select_wave / expand_wave / search)."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("TreeEngine needs a HIP device (no CPU path)")
        self.device, self.B, self.max_sims = dev, int(num_games), int(max_sims)
        B = self.B
        if float(reuse_factor) < 0:
            reuse_factor = auto_reuse_factor(B, self.max_sims, dev)
        self.reuse_factor = float(reuse_factor)
        extra = int(max(0.0, float(reuse_factor)) * self.max_sims)
        self.node_cap = self.max_sims + 2 + extra
        self.path_cap = min(self.node_cap + 1, PATH_CAP)
        if self.node_cap > MAX_NODE_CAP:
            raise ValueError(f"at most {MAX_NODE_CAP} nodes per game, got {self.node_cap}")
        self.edge_chunk = int(edge_chunk)
        if self.edge_chunk < 128 or self.edge_chunk & (self.edge_chunk - 1):
            raise ValueError(f"edge_chunk must be a power of two >= 128, got {edge_chunk}")
        self.chunk_cap = chunk_cap_for(self.node_cap, self.edge_chunk)
        self.pool_chunks = int(pool_chunks) if pool_chunks is not None else \
            auto_pool_chunks(B, self.max_sims, self.node_cap, self.edge_chunk, dev)
        if (self.pool_chunks + 1) * self.edge_chunk > (1 << 31):
            raise ValueError(f"edge pool of {self.pool_chunks} x {self.edge_chunk} records exceeds 2^31 pool indices")
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.buf: Dict[str, torch.Tensor] = {
            "root_state": z((B, 4), torch.int64),
            "nodes": z((B * self.node_cap, 6), torch.int64),            # 48-byte node records
            "edges": torch.empty((self.pool_chunks * self.edge_chunk, 4), dtype=torch.int64, device=dev),   # 32-byte records
            "n_nodes": z((B,), torch.int32), "n_edges": z((B,), torch.int32), "root_visits": z((B,), torch.int32),
            "root_w": z((B,), torch.float64), "root_init_value": z((B,), torch.float32),
            "path": z((B * self.path_cap,), torch.int32), "path_len": z((B,), torch.int32),
            "leaf_kind": z((B,), torch.int32), "leaf_state": z((B, 4), torch.int64), "leaf_value": z((B,), torch.float32),
            "root_terminal": z((B,), torch.uint8), "active": torch.ones((B,), dtype=torch.uint8, device=dev),
            "leaf_edge": z((B,), torch.int32), "leaf_parent": z((B,), torch.int32),
            "chunk_list": z((B * self.chunk_cap,), torch.int32), "n_chunks": z((B,), torch.int32),
            "free_chunks": torch.arange(self.pool_chunks, dtype=torch.int32, device=dev),
            "pool_top": torch.full((1,), self.pool_chunks, dtype=torch.int32, device=dev),
            # [0] expansions refused because the pool was empty (0 in a correctly sized engine), [1] fewest free chunks seen
            # + [2] fresh roots that still have to take the chunk of their first expansion (reserved for them)
            "pool_stats": torch.tensor([0, self.pool_chunks, 0], dtype=torch.int32, device=dev),
        }
        d = LzTreeDesc()
        d.num_games, d.node_cap, d.edge_chunk, d.path_cap = B, self.node_cap, self.edge_chunk, self.path_cap
        d.chunk_cap, d.pool_chunks = self.chunk_cap, self.pool_chunks
        d.exploration_weight = float(exploration_weight)
        for name, t in self.buf.items():
            setattr(d, name, t.data_ptr())
        self.desc = d
        self.batch_k = max(1, int(batch_k))
        if self.batch_k > 32:
            raise ValueError(f"batch_k must be <= 32, got {self.batch_k}")
        K = self.batch_k
        self.wbuf: Dict[str, torch.Tensor] = {}
        self.wdesc = None
        if K > 1:                                                       # slot-major [K][B] per-leaf arrays of a wave
            self.wbuf = {
                "path": z((K * B * WAVE_PATH_CAP,), torch.int32), "path_len": z((K * B,), torch.int32),
                "leaf_kind": z((K * B,), torch.int32), "leaf_state": z((K * B, 4), torch.int64),
                "leaf_value": z((K * B,), torch.float32), "leaf_edge": z((K * B,), torch.int32),
                "leaf_parent": z((K * B,), torch.int32), "sims_done": z((B,), torch.int32),
                "unfinished": z((1,), torch.int32), "eval_row": z((K * B,), torch.int32),
                "eval_state": z((K * B, 4), torch.int64), "eval_count": z((1,), torch.int64),
                "eval_total": z((1,), torch.int64),
            }
            wd = LzTreeWaveDesc()
            wd.batch_k, wd.path_cap = K, WAVE_PATH_CAP
            wd.max_backtrack_steps = max(0, int(max_backtrack_steps))
            for name, t in self.wbuf.items():
                setattr(wd, name, t.data_ptr())
            self.wdesc = wd
        # evaluator scratch (fused network inputs / outputs): one row per leaf slot of a wave
        self.planes = z((B, 11, 6, 6), torch.float32)
        self.lp1, self.lp2, self.lpm = (z((K * B, 36), torch.float32) for _ in range(3))
        self.values = z((K * B,), torch.float32)
        # finish outputs
        self.policy_dense = z((B, TOTAL_ACTION_DIM), torch.float32)
        self.chosen_index = z((B,), torch.int32)
        self.chosen_code = z((B, 4), torch.int32)
        self.chosen_valid = z((B,), torch.bool)
        self.terminal_mask = z((B,), torch.bool)
        self.root_value = z((B,), torch.float32)
        self.child_count = z((B,), torch.int32)
        self.child_action = z((B, OUT_CAP), torch.int32)
        self.child_visits = z((B, OUT_CAP), torch.int32)
        self.child_prior = z((B, OUT_CAP), torch.float32)
        self.reuse_dropped = z((2,), torch.int32)      # [0] kept subtrees dropped whole (defensive), [1] pruned to fit
        self.eval_count = z((B,), torch.int32)          # evaluations the games' expand steps consumed (LzTreeDesc.eval_count)
        self.desc.eval_count = self.eval_count.data_ptr()
        self.compact_evals = bool(compact_evals) and self.batch_k == 1
        self.live_total = z((1,), torch.int64)          # compact_evals: evaluations launched so far (sum of the per-simulation counts)
        if self.compact_evals:
            self.live = {"live_state": z((B, 4), torch.int64), "live_row": z((B,), torch.int32),
                         "live_count": z((self.max_sims + 2,), torch.int64)}
            for name, t in self.live.items():
                setattr(self.desc, name, t.data_ptr())
            self.desc.live_count_cap = self.max_sims + 2
        # persistent search kernel (lz_tree_search_persistent, one launch per move): built, parity-tested and MEASURED
        # SLOWER than the two-stream launch pairs at C2 (profiles/r03_experiments.md: 164-170 k against 192-194 k
        # positions/s), so it is opt-in (LZ_TREE_PERSISTENT=1 / the `persistent` attribute).  The CU's second workgroup
        # starts `stagger_us` late (LZ_TREE_STAGGER_US) so that the pair sharing a CU alternates network pass / tree step
        self.persistent = _persistent_default()
        self.stagger_us = int(os.environ.get("LZ_TREE_STAGGER_US", "40"))
        self._cu_slots: Optional[torch.Tensor] = None
        self.phase_ticks: Optional[torch.Tensor] = None

    def enable_trace(self, steps: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Parity tests: record, per step of a search (slot 0 = root step, slot s = s-th simulation), what the expand
        kernel consumed -- leaf kind / state, head rows, the softmax over the legal set, the value (LzTreeDesc.trace_*).
        Must be called before the search is captured into a graph (kernel arguments are frozen at capture)."""
        n = (self.max_sims + 1) if steps is None else int(steps)
        B, dev = self.B, self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.trace = {"trace_kind": z((n, B), torch.int32), "trace_leaf": z((n, B, 4), torch.int64),
                      "trace_heads": z((n, B, 108), torch.float32), "trace_priors": z((n, B, 220), torch.float32),
                      "trace_value": z((n, B), torch.float32)}
        for name, t in self.trace.items():
            setattr(self.desc, name, t.data_ptr())
        self.desc.trace_cap = n
        return self.trace

    _NODE_DT = [("w0", "<u8"), ("w1", "<u8"), ("w2", "<u8"), ("w3", "<u8"), ("edge_begin", "<i4"), ("nedges", "<i4"),
                ("parent", "<i4"), ("old_begin", "<i4")]
    _EDGE_DT = [("W", "<f8"), ("P", "<f4"), ("n_info", "<u4"), ("child", "<i4"), ("cbegin", "<i4"),
                ("act", "u1"), ("cn", "u1"), ("pad", "V6")]

    def game_nodes(self, g: int, count: Optional[int] = None):
        """Host copy of game g's node records (numpy structured array; inspection / tests, not the hot loop)."""
        import numpy as np
        nn = int(self.buf["n_nodes"][g]) if count is None else int(count)
        return self.buf["nodes"].view(self.B, self.node_cap, 6)[g, :nn].contiguous().cpu().numpy().view(np.dtype(self._NODE_DT)).reshape(nn)

    def edge_run(self, begin: int, count: int):
        """Host copy of `count` edge records from pool index `begin` (a node's run: Node.edge_begin / nedges)."""
        import numpy as np
        return self.buf["edges"][int(begin):int(begin) + int(count)].contiguous().cpu().numpy().view(np.dtype(self._EDGE_DT)).reshape(int(count))

    def root_edge_records(self, g: int):
        """The root's edge records of game g, read from the pool (host copy: inspection / adapters, not the hot loop):
        list of {action_index, prior, visit_count, value_sum (child mover's side), child_white, terminal}."""
        node = self.game_nodes(g, 1)[0]
        e0, ne = int(node["edge_begin"]), int(node["nedges"])
        if ne <= 0:
            return []
        recs = self.edge_run(e0, ne)
        return [{"action_index": int(r["act"]), "prior": float(r["P"]), "visit_count": int(r["n_info"] & 0xFFFFFF),
                 "value_sum": float(r["W"]), "child_white": bool((r["n_info"] >> 24) & 1),
                 "terminal": bool((r["n_info"] >> 24) & 2)} for r in recs]

    def pool_status(self) -> Dict[str, int]:
        """Edge pool: chunks, free now, fewest free seen since construction, refused expansions (must stay 0)."""
        refused, low = (int(v) for v in self.buf["pool_stats"][:2].tolist())
        return {"chunks": self.pool_chunks, "chunk_edges": self.edge_chunk, "free": int(self.buf["pool_top"].item()),
                "fewest_free": low, "refused_expansions": refused,
                "bytes": self.pool_chunks * self.edge_chunk * 32}

    def hbm_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.buf.values())

    def _stream(self):
        return L.stream_ptr(self.device)

    # ---- state plumbing ----
    def set_roots(self, state: GpuStateBatch, active: Optional[torch.Tensor] = None) -> None:
        ts = [t if t.is_contiguous() else t.contiguous() for t in state.tensors()]
        s = L.soa(ts)
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_pack_states(C.byref(s), L.i64(self.B), L.ptr(self.buf["root_state"]), self._stream()),
                    "pack_states")
        if active is not None:
            self.buf["active"].copy_(active.to(torch.uint8))
        else:
            self.buf["active"].fill_(1)

    def begin(self) -> None:
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_begin(C.byref(self.desc), self._stream()), "tree_begin")

    def select(self) -> None:
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_select(C.byref(self.desc), self._stream()), "tree_select")

    def leaf_planes(self) -> torch.Tensor:
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_packed_to_model_input(L.ptr(self.buf["leaf_state"]), L.i64(self.B), L.ptr(self.planes),
                                                     self._stream()), "packed_to_model_input")
        return self.planes

    def expand(self, *, is_root: bool, values: torch.Tensor, heads=None, priors220: Optional[torch.Tensor] = None,
               noise: Optional[torch.Tensor] = None, epsilon: float = 0.25) -> None:
        lp1 = lp2 = lpm = None
        if priors220 is None:
            lp1, lp2, lpm = heads
        nz_stride = int(noise.shape[1]) if noise is not None else 0
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_expand(C.byref(self.desc), C.c_int(1 if is_root else 0), L.ptr(lp1), L.ptr(lp2),
                                           L.ptr(lpm), L.ptr(priors220), L.ptr(values), L.ptr(noise), L.i64(nz_stride),
                                           C.c_float(float(epsilon)), self._stream()), "tree_expand")

    # ---- waves of batch_k leaves (legacy search, src/mcts.py:318-497) ----
    def select_wave(self, sims: int, reset_budget: bool = False) -> None:
        """Up to min(batch_k, sims - used) distinct leaves per game into the [batch_k][B] slot arrays (wbuf)."""
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_wave_select(C.byref(self.desc), C.byref(self.wdesc), L.i64(sims),
                                                C.c_int(1 if reset_budget else 0), self._stream()), "tree_wave_select")

    def expand_wave(self, *, values: torch.Tensor, heads=None, priors220: Optional[torch.Tensor] = None,
                    slot_major: bool = False) -> None:
        """Back up / expand the wave's leaves in the reference's order.  Evaluator rows: the compact list
        (wbuf eval_state / eval_row), or slot-major [batch_k][B] with `slot_major` / `priors220`."""
        lp1 = lp2 = lpm = None
        if priors220 is None:
            lp1, lp2, lpm = heads
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_wave_expand(C.byref(self.desc), C.byref(self.wdesc), L.ptr(lp1), L.ptr(lp2),
                                                L.ptr(lpm), L.ptr(priors220), L.ptr(values),
                                                C.c_int(1 if slot_major else 0), self._stream()),
                    "tree_wave_expand")

    def search_waves(self, net: FusedNet, sims: int, waves: int, noise: Optional[torch.Tensor] = None,
                     epsilon: float = 0.25, continue_trees: bool = False, skip_roots: bool = False) -> None:
        """[roots ->] `waves` x (select batch_k leaves, evaluate batch_k * B slots, expand + backup); C++ loop."""
        nz_stride = int(noise.shape[1]) if noise is not None else 0
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_search_waves(
                C.byref(self.desc), C.byref(self.wdesc), C.byref(net.desc), L.i64(sims), L.i64(waves), L.ptr(self.lp1),
                L.ptr(self.lp2), L.ptr(self.lpm), L.ptr(self.values), L.ptr(noise), L.i64(nz_stride),
                C.c_float(float(epsilon)), C.c_int(1 if continue_trees else 0), C.c_int(1 if skip_roots else 0),
                self._stream()), "tree_search_waves")

    def finish_waves(self, net: FusedNet, sims: int, max_rounds: int = 4096) -> int:
        """Games that found fewer open leaves than batch_k in some wave still have budget: further rounds until every
        game has used `sims` simulations (one host read per round; rare outside tiny endgame trees)."""
        rounds, chunk = 0, 8             # a round in which every game is done costs a few empty launches: 8 per host read
        while int(self.wbuf["unfinished"].item()) > 0 and rounds < max_rounds:
            self.search_waves(net, sims, chunk, skip_roots=True)
            rounds += chunk
        return rounds

    def finish(self, temperatures: torch.Tensor, uniforms: Optional[torch.Tensor],
               target_temperatures: Optional[torch.Tensor] = None, prior_pseudocount: float = 0.0,
               force_uniform: Optional[torch.Tensor] = None, sample_moves: Optional[bool] = None) -> None:
        """`sample_moves` defaults to "sample iff uniforms are given"."""
        sample = uniforms is not None if sample_moves is None else bool(sample_moves)
        if force_uniform is not None and force_uniform.dtype != torch.uint8:
            force_uniform = force_uniform.to(torch.uint8)
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_finish(C.byref(self.desc), L.ptr(temperatures), L.ptr(target_temperatures),
                                           C.c_float(float(prior_pseudocount)), L.ptr(force_uniform), C.c_int(int(sample)),
                                           L.ptr(uniforms),
                                           L.ptr(self.policy_dense), L.ptr(self.chosen_index), L.ptr(self.chosen_code),
                                           L.ptr(self.chosen_valid), L.ptr(self.terminal_mask), L.ptr(self.root_value),
                                           L.ptr(self.child_count), L.ptr(self.child_action), L.ptr(self.child_visits),
                                           L.ptr(self.child_prior), L.i64(OUT_CAP), self._stream()), "tree_finish")

    def advance(self, played_action: Optional[torch.Tensor] = None, reset: Optional[torch.Tensor] = None,
                next_sims: Optional[int] = None) -> None:
        """Tree reuse (a21): promote the played child of every game to root, keeping its subtree.  Call after
        set_roots() with the post-move states; `played_action` int32[B] defaults to the last finish()'s picks."""
        pa = self.chosen_index if played_action is None else played_action.to(torch.int32).contiguous()
        if reset is not None and reset.dtype != torch.uint8:
            reset = reset.to(torch.uint8)
        sims = self.max_sims if next_sims is None else int(next_sims)
        with torch.cuda.device(self.device):
            L.check(L.lib().lz_tree_advance(C.byref(self.desc), L.ptr(pa), L.ptr(reset), L.i64(sims),
                                            L.ptr(self.reuse_dropped[0:1]), L.ptr(self.reuse_dropped[1:2]),
                                            self._stream()), "tree_advance")

    def search(self, net: FusedNet, sims: int, noise: Optional[torch.Tensor] = None, epsilon: float = 0.25,
               continue_trees: bool = False, compact: Optional[bool] = None) -> None:
        """Whole search of one move (C++ loop: begin, root, sims x select/eval/expand); `continue_trees`: the
        trees were prepared by advance() (kept subtrees / fresh roots), no begin.  `compact` (engines built with
        `compact_evals`): False launches every slot of every simulation (cheaper while nearly all games are live: no scan
        kernel per simulation), True / None the compact lists; results are bit-identical either way."""
        if int(sims) > self.max_sims:
            raise ValueError(f"sims={sims} exceeds the arena capacity max_sims={self.max_sims}")
        if self.batch_k > 1:
            self.search_waves(net, sims, -(-int(sims) // self.batch_k), noise, epsilon, continue_trees)
            return
        nz_stride = int(noise.shape[1]) if noise is not None else 0
        if self.persistent_ok(net):
            # one launch per move: a workgroup owns 8 games, no kernel boundary between simulations (csrc/lz_search.hip)
            if self._cu_slots is None:
                self._cu_slots = torch.zeros((4096,), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.lib().lz_tree_search_persistent(
                    C.byref(self.desc), C.byref(net.desc), L.i64(sims), L.ptr(self.lp1), L.ptr(self.lp2), L.ptr(self.lpm),
                    L.ptr(self.values), L.ptr(noise), L.i64(nz_stride), C.c_float(float(epsilon)),
                    C.c_int(1 if continue_trees else 0), L.ptr(self._cu_slots), L.i64(self.stagger_us),
                    L.ptr(self.phase_ticks), self._stream())
            if rc in (-2, -3) and not torch.cuda.is_current_stream_capturing():
                # LZ_ERR_UNSUPPORTED / LZ_ERR_LAUNCH (e.g. the kernel's ~80 KB of dynamic LDS refused on this part): the
                # header promises the launch-pair search as the fallback -- nothing has run yet, so take it, for good
                print(f"[liuzhou_amd] persistent search kernel unavailable (status {rc}); using lz_tree_search", flush=True)
                self.persistent = False
            else:
                L.check(rc, "tree_search_persistent")
                return
        fn = L.lib().lz_tree_search_continue if continue_trees else L.lib().lz_tree_search
        use_lists = self.compact_evals and (compact is None or bool(compact))
        desc = self.desc
        if self.compact_evals and not use_lists:                    # the same engine, dense launches for this move
            desc = LzTreeDesc()
            C.memmove(C.byref(desc), C.byref(self.desc), C.sizeof(LzTreeDesc))
            desc.live_state = desc.live_row = desc.live_count = None
            desc.live_count_cap = 0
        with torch.cuda.device(self.device):
            L.check(fn(C.byref(desc), C.byref(net.desc), L.i64(sims), L.ptr(self.planes),
                       L.ptr(self.lp1), L.ptr(self.lp2), L.ptr(self.lpm), L.ptr(self.values),
                       L.ptr(noise), L.i64(nz_stride), C.c_float(float(epsilon)), self._stream()),
                    "tree_search")
        if self.compact_evals:                                      # evaluations launched: the leaves the lists held / every slot
            if use_lists:
                self.live_total.add_(self.live["live_count"][: int(sims) + 1].sum())
            else:
                self.live_total.add_(self.B * (int(sims) + 1))

    def persistent_ok(self, net: FusedNet) -> bool:
        """Whether search() runs the one-launch-per-move kernel: 64-channel net in fp16 mode, not switched off
        (`persistent` attribute; env LZ_TREE_PERSISTENT=0)."""
        return (bool(self.persistent) and self.batch_k <= 1 and int(net.desc.channels) == 64 and not (int(net.desc.flags) & (4 | 8))
                and not self.compact_evals)

    def enable_phase_ticks(self) -> torch.Tensor:
        """Measurement aid: int64[workgroups, 4] = 100 MHz ticks each workgroup of the persistent search kernel spent in
        network passes / tree steps during the LAST search, its arrival slot on its CU and the CU key (must be enabled
        before the search is captured)."""
        n = int(L.lib().lz_tree_search_persistent_grid(L.i64(self.B)))
        self.phase_ticks = torch.zeros((n, 4), dtype=torch.int64, device=self.device)
        return self.phase_ticks


def _persistent_default() -> bool:
    return os.environ.get("LZ_TREE_PERSISTENT", "0").strip().lower() in ("1", "on", "true")


def persistent_search_available(net, batch_k: int = 1) -> bool:
    """Does the search of `net` run as the one-launch-per-move kernel (csrc/lz_search.hip; opt-in)?  Then one engine over
    all games already puts two workgroups on every CU, and the two-stream split (DualStreamTreeMCTS) has nothing to add."""
    if not _persistent_default():
        return False
    return isinstance(net, FusedNet) and int(batch_k) <= 1 and int(net.desc.channels) == 64 and not (int(net.desc.flags) & 4)


def dirichlet_noise(shape, alpha: float, device, generator=None) -> torch.Tensor:
    """Gamma(alpha) draws; the expand kernel mixes them with weight epsilon and renormalises the priors
    (the same distribution as Dirichlet noise after normalisation over the legal children)."""
    a = torch.full(shape, float(alpha), dtype=torch.float32, device=device)
    g = torch._standard_gamma(a) if generator is None else torch._standard_gamma(a, generator=generator)
    return g


class PriorEvaluator:
    """External evaluator of the split-phase protocol in the reference's own form (`PortableCppTreeBatch`:
    prepare roots -> evaluate -> complete(priors, values), v1/python/portable_cpp_mcts.py:270-282):
    `fn(planes float32[B,11,6,6], packed_states int64[B,4]) -> (priors float32[B,220], values float32[B])`, both on the
    engine's device; the priors are used as given (renormalised over the legal set by the expand kernel)."""

    def __init__(self, fn) -> None:
        self.fn = fn

    def __call__(self, planes: torch.Tensor, packed: torch.Tensor):
        return self.fn(planes, packed)


class PortableTreeMCTS:
    """Full-tree search with the V1RootMCTS calling convention (search_batch -> RootSearchBatchOutput)."""

    def __init__(self, net: FusedNet, num_games: int, num_simulations: int, device, exploration_weight: float = 1.0,
                 add_dirichlet_noise: bool = True, dirichlet_alpha: float = 0.3, dirichlet_epsilon: float = 0.25,
                 sample_moves: bool = True, use_graph: Optional[bool] = None, reuse_tree: bool = False,
                 reuse_factor: float = -1.0, policy_target_temperature: Optional[float] = None,
                 policy_target_prior_pseudocount: float = 0.0, batch_k: int = 1, seed: int = 12345,
                 game_offset: int = 0, game_stride: Optional[int] = None, trace: bool = False,
                 collect_timing: bool = False, compact_evals: Optional[bool] = None) -> None:
        """`compact_evals` (fused network only): per simulation only the leaves that need the network are evaluated
        (`TreeEngine(compact_evals=True)`); True = in every search; None = the engine is built for both forms when a launch
        of all slots is at least two network passes per CU (C3: eight) and uses the lists for the searches that say so
        (`search_batch(compact=True)`, the runner's hint that the wave is draining) -- env LZ_TREE_COMPACT=0 / 1 overrides.
        `net`: a FusedNet (the production path: the whole search of a move is enqueued from C++ with the fused network
        kernel in the loop, one hipGraph per move), or any module returning `ChessNet.forward`'s 4-tuple -- then the search
        runs the split-phase protocol (select -> planes -> module -> expand) with that module as an external fp32
        evaluator, on its own device (the reference's `PortableMCTS` with an arbitrary model, portable_mcts.py:381-386).
        `seed` / `game_offset` / `game_stride`: keys of the per-game counter RNG (game_rng.GameRng) behind the root noise
        and the move sampling.  `trace`: record what every expand step consumed (TreeEngine.enable_trace; parity tests).
        `batch_k` > 1: the legacy search's waves (src/mcts.py `batch_K`, default 16 there): up to batch_k distinct
        leaves per game are collected, evaluated in one network launch of batch_k * B positions and backed up per wave.
        `reuse_tree`: keep the played child's subtree between consecutive search_batch calls on the same games
        (the reference's portable self-play does, v1/python/portable_self_play.py:191); the arenas then hold
        (1 + reuse_factor) * sims nodes per game (reuse_factor < 0: as much as a third of the free memory allows).  `policy_target_*`: portable_mcts.py:690-700."""
        self.net, self.sims = net, int(num_simulations)
        self.fused = isinstance(net, FusedNet)
        self.reuse_tree = bool(reuse_tree)
        self.batch_k = max(1, int(batch_k))
        env_c = os.environ.get("LZ_TREE_COMPACT", "").strip()
        if env_c in ("0", "1"):
            compact_evals = env_c == "1"
        # asked for explicitly: every search uses the lists.  Chosen automatically: the engine can do both and launches
        # every slot unless a search is told otherwise (`compact=` of search_batch: the runner's hint that the wave is
        # draining) -- while nearly all games are live the lists shorten nothing and cost a scan kernel per simulation
        self.compact_default = compact_evals is True
        if compact_evals is None:
            compact_evals = self.fused and int(num_games) >= 2 * samples_per_launch_pass(net)
        self.engine = TreeEngine(num_games, num_simulations, device, exploration_weight,
                                 reuse_factor=float(reuse_factor) if self.reuse_tree else 0.0, batch_k=self.batch_k,
                                 compact_evals=bool(compact_evals) and self.fused and self.batch_k == 1)
        if trace:
            self.engine.enable_trace()
        if not self.fused:
            if self.batch_k > 1:
                raise ValueError("an external evaluator supports batch_k = 1 only")
            self._eval_device = None if isinstance(net, PriorEvaluator) else next(net.parameters()).device
        self.rng = GameRng(num_games, device, seed=seed, game_offset=game_offset, game_stride=game_stride)
        self._game0 = self.rng.game.clone()
        self._uniforms = torch.zeros((self.engine.B,), dtype=torch.float32, device=self.engine.device)
        self.injected_noise: Optional[torch.Tensor] = None       # parity runs: [B, <= OUT_CAP] instead of the RNG's draws
        self.injected_uniforms: Optional[torch.Tensor] = None
        self._evals_dev = torch.zeros((1,), dtype=torch.int64, device=self.engine.device)
        self.extra_rounds = 0
        self.graph_retry_off = False                             # a failed graph capture made this engine launch directly
        self._collect_timing = bool(collect_timing)
        self._timing_events = []
        self._timing_ms: Dict[str, float] = {}
        self._timing_calls: Dict[str, int] = {}
        self.target_temperature = None if policy_target_temperature is None else float(policy_target_temperature)
        self.prior_pseudocount = float(policy_target_prior_pseudocount)
        self._have_trees = False
        self.add_noise, self.alpha, self.eps = bool(add_dirichlet_noise), float(dirichlet_alpha), float(dirichlet_epsilon)
        self.sample_moves = bool(sample_moves)
        self._root_evals = 0
        # hipGraph: the whole search of a move (1 + sims) x (select, fused net, expand) is captured once and
        # replayed every move; all buffers are static, noise / roots are refreshed in place before the replay.
        if use_graph is None:
            use_graph = os.environ.get("LZ_TREE_GRAPH", "on").strip().lower() not in ("off", "0", "false")
        self.use_graph = bool(use_graph)
        self._graphs = {}
        self._tail_graph = None
        self._tail_rounds = 0
        self._noise_buf = torch.zeros((self.engine.B, OUT_CAP), dtype=torch.float32, device=self.engine.device)

    # ---- tracing: roctx ranges (torch.cuda.nvtx is roctx on ROCm) + HIP-event timing buckets, as the reference's
    # V1RootMCTS._nvtx_range / _timed (v1/python/mcts_gpu.py:565-602); bucket names follow its runner
    # (self_play_gpu_runner.py:276-281): root_puct_ms = the search, pack_writeback_ms = policy / pick extraction ----
    @contextmanager
    def _timed(self, name: str):
        torch.cuda.nvtx.range_push(f"lz.tree.{name}")
        if not self._collect_timing:
            try:
                yield
            finally:
                torch.cuda.nvtx.range_pop()
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(self.engine.device))
        try:
            yield
        finally:
            b.record(torch.cuda.current_stream(self.engine.device))
            self._timing_events.append((name, a, b))
            torch.cuda.nvtx.range_pop()

    def get_timing(self, reset: bool = False) -> Dict[str, Dict[str, float]]:
        """{"timing_ms": {...}, "timing_calls": {...}} like V1RootMCTS.get_timing (one device synchronisation)."""
        if self._timing_events:
            torch.cuda.synchronize(self.engine.device)
            for name, a, b in self._timing_events:
                self._timing_ms[name] = self._timing_ms.get(name, 0.0) + float(a.elapsed_time(b))
                self._timing_calls[name] = self._timing_calls.get(name, 0) + 1
            self._timing_events = []
        out = {"timing_ms": dict(self._timing_ms), "timing_calls": dict(self._timing_calls)}
        if reset:
            self._timing_ms, self._timing_calls = {}, {}
        return out

    def _evaluate_external(self, planes: torch.Tensor):
        """External fp32 evaluator: module(planes) -> (log_p1, log_p2, log_pmc, value_logits); scalar value = bucket
        expectation (src/neural_network.py:201-210), everything back on the engine's device."""
        dev = self.engine.device
        with torch.inference_mode():
            lp1, lp2, lpm, raw = self.net(planes.to(self._eval_device))
            probs = torch.softmax(raw.float(), dim=-1)
            centers = torch.linspace(-1.0, 1.0, steps=raw.shape[-1], dtype=probs.dtype, device=probs.device)
            val = (probs * centers).sum(-1)
        f = lambda t: t.float().to(dev).contiguous()
        return f(lp1), f(lp2), f(lpm), f(val)

    def _search_split(self, noise: Optional[torch.Tensor], continue_trees: bool) -> None:
        e = self.engine
        if not continue_trees:
            e.begin()
        for s in range(self.sims + 1):
            if s > 0:
                e.select()
            if isinstance(self.net, PriorEvaluator):
                pri, val = self.net(e.leaf_planes(), e.buf["leaf_state"])
                e.expand(is_root=(s == 0), values=val.to(torch.float32).contiguous(),
                         priors220=pri.to(torch.float32).contiguous(), noise=noise if s == 0 else None, epsilon=self.eps)
                continue
            lp1, lp2, lpm, val = self._evaluate_external(e.leaf_planes())
            e.expand(is_root=(s == 0), values=val, heads=(lp1, lp2, lpm), noise=noise if s == 0 else None,
                     epsilon=self.eps)

    def _search(self, add_noise: bool, continue_trees: bool) -> None:
        e = self.engine
        noise = self._noise_buf if add_noise else None
        if not self.fused:
            self._search_split(noise, continue_trees)
            return
        lists = bool(getattr(self, "compact_now", self.compact_default)) and e.compact_evals
        self.list_searches = getattr(self, "list_searches", 0) + int(lists)      # searches that used the compact lists
        self.last_search_lists = lists
        if not self.use_graph:
            e.search(self.net, self.sims, noise, self.eps, continue_trees, compact=lists)
            return
        key = (add_noise, continue_trees, lists)
        g = self._graphs.get(key)
        if g is None:
            self.captures = getattr(self, "captures", 0) + 1        # this search synchronises the host (warm-up, capture)
            if not continue_trees:
                # warm-up launch outside capture: every kernel of the search must be loaded before a stream capture
                # starts.  Two simulations reach all of them (root expand + select, expand + select, last expand), and a
                # fresh search starts by resetting the trees, so what the warm-up leaves behind does not matter -- a full
                # search here cost 1.6 s per engine at C3.  A continued search must run exactly once and is only reached
                # after a fresh one
                # (an engine with compact lists warms up both launch forms: the other one may be captured later inside a
                #  continued search, which cannot be preceded by a warm-up of its own)
                counted = (e.live_total.clone(), e.eval_count.clone())     # warm-ups are not searches: not counted
                for form in ((lists, not lists) if e.compact_evals else (lists,)):
                    e.search(self.net, min(self.sims, 2), noise, self.eps, False, compact=form)
                e.live_total.copy_(counted[0]); e.eval_count.copy_(counted[1])
            torch.cuda.synchronize(e.device)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    e.search(self.net, self.sims, noise, self.eps, continue_trees, compact=lists)
                if os.environ.get("LZ_TREE_GRAPH_FAULT", "") == "capture":      # test hook: pretend the capture failed
                    raise RuntimeError("injected stream capture failure (LZ_TREE_GRAPH_FAULT=capture)")
            except RuntimeError as exc:
                # Capture failed (the reference's worker retries a failed finalize-graph capture once with the graph off,
                # v1/python/self_play_worker.py:24-30,434-442): nothing of the captured search has run, so the same search
                # is launched directly, and this engine stays on direct launches.  Recorded in the manifest.
                torch.cuda.synchronize(e.device)
                self.use_graph = False
                self.graph_retry_off = True
                self._graphs.clear()
                print(f"[liuzhou_amd] hipGraph capture of the tree search failed ({exc!r}); falling back to direct launches",
                      flush=True)
                e.search(self.net, self.sims, noise, self.eps, continue_trees, compact=lists)
                return
            self._graphs[key] = g
        g.replay()

    @property
    def leaf_evals(self) -> int:
        """Network evaluations so far (batch_k > 1: one host read of the device-side counters)."""
        if self.engine.compact_evals:
            return int(self.engine.live_total.item())
        n = self._root_evals
        if self.batch_k > 1:
            n += int(self.engine.wbuf["eval_total"].item()) + int(self.engine.wbuf["eval_count"].item())
        return n

    @property
    def consumed_evals(self) -> int:
        """Evaluations the searches actually consumed (expanded leaves and fresh roots): `leaf_evals` minus terminal
        leaves, inactive slots and the unused evaluation of kept roots.  One host read."""
        return int(self.engine.eval_count.sum(dtype=torch.int64).item())

    def _finish_waves(self) -> None:
        """batch_k > 1: games whose waves found fewer open leaves than batch_k still have budget (narrow trees)."""
        if self.batch_k <= 1:
            return
        e = self.engine
        while self.tail_step():
            pass

    def tail_step(self) -> bool:
        """One host read of `unfinished` (waits for this engine's stream only) and, if some game still has budget, 8
        more rounds replayed as one graph (the leftover rounds are a handful of games with tiny network batches).
        Returns whether rounds were launched -- several engines on several streams take turns calling this."""
        e = self.engine
        if self.batch_k <= 1 or int(e.wbuf["unfinished"].item()) == 0 or self._tail_rounds >= 4096:
            self._tail_rounds = 0
            return False
        if not self.use_graph:
            e.search_waves(self.net, self.sims, 8, skip_roots=True)
        else:
            if self._tail_graph is None:
                torch.cuda.synchronize(e.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    e.search_waves(self.net, self.sims, 8, skip_roots=True)
                self._tail_graph = g
            self._tail_graph.replay()
        self._tail_rounds += 8
        self.extra_rounds += 8
        return True

    def reset_trees(self) -> None:
        """Forget the kept subtrees: the next search_batch starts every game from a fresh root."""
        self._have_trees = False

    def reset_run(self, seed: int) -> None:
        """A cached engine starts another run (self_play_tree_gpu called again with the same network and shape): fresh
        trees, RNG keys and counters as after construction; arenas, descriptors and captured graphs are kept."""
        self._have_trees = False
        self.rng.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.rng.game.copy_(self._game0)
        self.rng.ply.zero_()
        self._root_evals = 0
        self.extra_rounds = 0
        self.engine.eval_count.zero_()
        self.engine.live_total.zero_()
        self.engine.reuse_dropped.zero_()
        # refused expansions / fewest free chunks are per run too (a cached engine must not report an earlier run's)
        # (pool_stats[2], the chunks reserved for fresh roots, follows the trees' state and is NOT a per-run statistic)
        self.engine.buf["pool_stats"][:2].copy_(torch.tensor([0, self.engine.pool_chunks], dtype=torch.int32))
        self.get_timing(reset=True)
        if self.batch_k > 1:
            self.engine.wbuf["eval_total"].zero_(); self.engine.wbuf["eval_count"].zero_()

    def prepare(self, state: GpuStateBatch) -> None:
        """Set-up outside any timed region: load every kernel and capture the graphs (a fresh search, and a continued
        one with subtree reuse) by searching `state` once or twice; the trees built here are thrown away."""
        temps = torch.ones((self.engine.B,), dtype=torch.float32, device=self.engine.device)
        game0 = self.rng.game.clone()
        for _ in range(2 if self.reuse_tree else 1):
            self.search_batch(state, temperatures=temps)
        self.rng.game.copy_(game0)
        self.rng.ply.zero_()
        self._have_trees = False
        self._root_evals = 0
        self.extra_rounds = 0
        self.engine.eval_count.zero_()
        self.engine.live_total.zero_()
        self.get_timing(reset=True)
        if self.batch_k > 1:
            self.engine.wbuf["eval_total"].zero_(); self.engine.wbuf["eval_count"].zero_()

    def search_batch(self, state: GpuStateBatch, *, temperatures: torch.Tensor, active: Optional[torch.Tensor] = None,
                     add_dirichlet_noise: Optional[bool] = None, reset: Optional[torch.Tensor] = None,
                     played_action: Optional[torch.Tensor] = None,
                     force_uniform_random_mask: Optional[torch.Tensor] = None,
                     rng_game_ids: Optional[torch.Tensor] = None,
                     rng_plies: Optional[torch.Tensor] = None, compact: Optional[bool] = None) -> RootSearchBatchOutput:
        """`compact`: see `launch_search`.
        `state`: the games' current positions.  With `reuse_tree`, games whose position is the child reached by
        `played_action` (default: the move this engine picked last time) keep that child's subtree; `reset` marks
        games that were re-seated.  Anything that does not match simply starts a fresh tree.  `rng_game_ids` /
        `rng_plies` int64[B]: the runner's own game numbering and ply counters as RNG keys (default: slot-derived ids
        that advance on `reset`, plies counted per search)."""
        self.launch_search(state, active=active, add_dirichlet_noise=add_dirichlet_noise, reset=reset,
                           played_action=played_action, rng_game_ids=rng_game_ids, rng_plies=rng_plies, compact=compact)
        return self.complete_search(state, temperatures=temperatures,
                                    force_uniform_random_mask=force_uniform_random_mask)

    def launch_search(self, state: GpuStateBatch, *, active: Optional[torch.Tensor] = None,
                      add_dirichlet_noise: Optional[bool] = None, reset: Optional[torch.Tensor] = None,
                      played_action: Optional[torch.Tensor] = None, rng_game_ids: Optional[torch.Tensor] = None,
                      rng_plies: Optional[torch.Tensor] = None, compact: Optional[bool] = None) -> None:
        """First half of search_batch: everything up to the end of the captured search, nothing that waits for the
        device (so that several engines on several streams can be launched back to back)."""
        e = self.engine
        # compact lists or dense launches for this move (engines built with compact_evals; the runner passes a hint from
        # the number of live games: the lists pay once a launch gets at least one network pass per CU shorter)
        self.compact_now = self.compact_default if compact is None else bool(compact)
        add_noise = self.add_noise if add_dirichlet_noise is None else bool(add_dirichlet_noise)
        e.set_roots(state, active)
        continue_trees = self.reuse_tree and self._have_trees
        self.rng.begin_move(reset if self._have_trees else None, rng_game_ids, rng_plies)
        self._explicit_plies = rng_plies is not None
        with self._timed("root_puct_ms"):
            if continue_trees:
                e.advance(played_action, reset, self.sims)
            if add_noise:
                if self.injected_noise is not None:
                    self._noise_buf.zero_()
                    self._noise_buf[:, : int(self.injected_noise.shape[1])].copy_(self.injected_noise.to(torch.float32))
                else:
                    self.rng.gamma_into(self._noise_buf, self.alpha, MAX_CHILDREN)
            self._search(add_noise, continue_trees)
        self._have_trees = True

    def complete_search(self, state: GpuStateBatch, *, temperatures: torch.Tensor,
                        force_uniform_random_mask: Optional[torch.Tensor] = None,
                        tails_done: bool = False) -> RootSearchBatchOutput:
        """Second half: leftover waves (batch_k > 1, host reads), policy / pick extraction, output record."""
        e = self.engine
        dev = e.device
        if self.batch_k > 1:
            if not tails_done:
                self._finish_waves()
            self._root_evals += e.B                                   # wave evaluations are counted on the device
        else:
            self._root_evals += e.B * (self.sims + 1)
        need_u = self.sample_moves or force_uniform_random_mask is not None
        uniforms = None
        with self._timed("pack_writeback_ms"):
            if need_u:
                uniforms = self._uniforms
                if self.injected_uniforms is not None:
                    uniforms.copy_(self.injected_uniforms.to(torch.float32))
                else:
                    self.rng.uniform_into(uniforms, PURPOSE_PICK)
            self.rng.end_move(getattr(self, "_explicit_plies", False))
            temps = temperatures.to(torch.float32).contiguous()
            tt = None if self.target_temperature is None else torch.full_like(temps, self.target_temperature)
            e.finish(temps, uniforms, tt, self.prior_pseudocount, force_uniform_random_mask, self.sample_moves)
            model_input = states_to_model_input(state)
            legal_mask, _ = encode_actions_fast(state)
        return RootSearchBatchOutput(
            model_input=model_input, legal_mask=legal_mask, policy_dense=e.policy_dense, root_value=e.root_value,
            terminal_mask=e.terminal_mask, chosen_action_indices=e.chosen_index.to(torch.int64),
            chosen_action_codes=e.chosen_code, chosen_valid_mask=e.chosen_valid)


class DualStreamTreeMCTS:
    """Two half-size searches on two HIP streams, same interface as PortableTreeMCTS.

    Within one game the simulation chain (select -> evaluate -> expand) is strictly sequential, and the tree kernel
    is latency-bound: while it runs, the matrix pipes idle.  With the games split in two halves, each with its own
    engine, hipGraph and stream, one half's tree kernel overlaps the other half's network kernel.  The network runs
    4-wave workgroups of 8 samples (two per CU, `FusedNet(half_workgroups=True)`), so that a half batch alone still
    covers every CU.  Results are identical to one engine over all games (games are independent)."""

    def __init__(self, model, num_games: int, num_simulations: int, device, num_parts: Optional[int] = None, **kw) -> None:
        dev = torch.device(device)
        self.B = int(num_games)
        k = 2 if num_parts is None else int(num_parts)      # 3 and 4 parts measured 34 % slower than 2 on C2
        k = max(1, min(k, self.B))
        base_n, extra = divmod(self.B, k)
        sizes = [base_n + (1 if i < extra else 0) for i in range(k)]
        self.bounds = []
        start = 0
        for n in sizes:
            self.bounds.append((start, start + n))
            start += n
        self.device = dev
        from .streams import overlapping_streams
        self.streams = overlapping_streams(dev, k)              # probed: two new streams CAN share a hardware queue
        self._pair_mode = overlapping_streams.last_mode
        self.parts = []
        base = model if isinstance(model, FusedNet) else FusedNet(model, dev)
        kw.pop("game_offset", None); kw.pop("game_stride", None)
        # one arena factor for the whole batch: sized per part, the first part would take its share of the free memory
        # and the next one a share of what is left -- unequal arenas, unequal subtree-drop rates
        if kw.get("reuse_tree", False) and float(kw.get("reuse_factor", -1.0)) < 0:
            kw["reuse_factor"] = auto_reuse_factor(self.B, int(num_simulations), dev)
        for (a, _b), n in zip(self.bounds, sizes):
            # waves of batch_k leaves are large launches: full 8-wave workgroups (they fill the chip on their own)
            half = int(kw.get("batch_k", 1)) <= 1
            # global game ids: slot a + g of the whole batch, so that a game's RNG stream does not depend on the split
            self.parts.append(PortableTreeMCTS(base.variant(half_workgroups=half), n, num_simulations, dev,
                                               game_offset=a, game_stride=self.B, **kw))
        self.sims = int(num_simulations)
        self.serialize = False        # measurement aid: run the halves one after the other on the caller's stream
        # Overlap watch.  A probed pair of streams can still end up on one hardware queue later (the runtime maps streams
        # to its 4 queues when they become active; other streams of the process come and go): the halves then run one
        # after the other, +50 % per ply, for the rest of the run (seen in ~1 of 6 worker runs inside a long-lived
        # process).  So the first searches, and every 64th after them, are bracketed by events; a search whose two halves
        # did not overlap (union of the two intervals ~ their sum) twice in a row makes the engine draw new streams.
        self._watch: List[List[torch.cuda.Event]] = []
        self._watch_left = 6
        self._searches = 0
        self._serial_seen = 0
        self.stream_redraws = 0
        # Round 6: a THIRD way of not sharing the chip showed up (a (-1, 0) priority pair in bench.py's C2 runner leg, every
        # time): both halves start together, take the same time -- and that time is what ONE half after the other takes
        # (31 ms instead of 20.5: their kernels alternate instead of running side by side).  Neither "union ~ sum" nor
        # "one nested in the other" sees that; only a reference does: one search of every run is launched with both halves
        # on ONE stream on purpose (`_t_serial` = what no overlap costs), and a search whose union exceeds 0.85 of it has
        # not overlapped, whatever its intervals look like (healthy: 0.66).
        self._t_serial: Optional[float] = None
        self.overlap_ratio: Optional[float] = None          # union / _t_serial of the last watched search (statistics)

    @property
    def leaf_evals(self) -> int:
        return sum(p.leaf_evals for p in self.parts)

    @property
    def consumed_evals(self) -> int:
        return sum(p.consumed_evals for p in self.parts)

    def prepare(self, state: GpuStateBatch) -> None:
        for (a, b), part in zip(self.bounds, self.parts):
            part.prepare(state._map(lambda t, a=a, b=b: t[a:b]))
        torch.cuda.synchronize(self.device)

    @property
    def use_graph(self) -> bool:
        return self.parts[0].use_graph

    @use_graph.setter
    def use_graph(self, v: bool) -> None:
        for p in self.parts:
            p.use_graph = bool(v)

    def reset_trees(self) -> None:
        for p in self.parts:
            p.reset_trees()

    def reset_run(self, seed: int) -> None:
        for p in self.parts:
            p.reset_run(seed)
        self._watch, self._watch_left, self._searches, self._serial_seen = [], 6, 0, 0
        self._t_serial = None                                # re-calibrated per run (the launch form may differ)

    def _check_overlap(self) -> None:
        """Look at finished overlap brackets (never waits) and re-draw the streams if the halves ran one after the other."""
        self._searches += 1
        if self._watch_left <= 0 and self._searches % 64 == 0:
            self._watch_left = 2
        while self._watch and all(e.query() for e in self._watch[0][:4]):
            s0, e0, s1, e1, calib = self._watch.pop(0)
            d0, d1 = s0.elapsed_time(e0), s1.elapsed_time(e1)
            if calib:                                            # the deliberately serialised search: the reference
                self._t_serial = d0 + d1
                continue
            self._watch_left -= 1
            lead = s0.elapsed_time(s1)                           # start of part 1 relative to part 0 (ms, may be < 0)
            union = max(d0, lead + d1) - min(0.0, lead)
            # Two signatures of halves that do not share the chip: (a) one after the other on one hardware queue -- disjoint
            # intervals, union ~ sum; (b) one half STARVED by the other -- nested intervals: both start together, one
            # finishes in about half the time of the other (seen once in round 6 with a (-1, 0) priority pair: the whole
            # runner leg at 31.9 ms per ply, which test (a) alone did not flag).  Halves that overlap properly take about the
            # same time: they are equal work on equal shares of the chip.
            serial = union > 0.9 * (d0 + d1) or min(d0, d1) < 0.62 * max(d0, d1)
            if self._t_serial is not None and self._t_serial > 2.0:
                # (c) whatever the intervals look like: the search took what the two halves take one after the other
                self.overlap_ratio = union / self._t_serial
                serial = serial or union > 0.85 * self._t_serial
            if os.environ.get("LZ_WATCH_DEBUG", "0") == "1":
                print(f"[lz watch] search {self._searches}: d0 {d0:.2f} ms, d1 {d1:.2f} ms, lead {lead:.2f} ms, union {union:.2f}, "
                      f"serial reference {self._t_serial} -> {'serial' if serial else 'ok'}", file=sys.stderr, flush=True)
            if d0 > 1.0 and d1 > 1.0 and serial:
                self._serial_seen += 1
                self._watch_left = max(self._watch_left, 2)
                if self._serial_seen >= 2 and self.stream_redraws < 6:
                    from .streams import overlapping_streams
                    self._old_streams = getattr(self, "_old_streams", []) + list(self.streams)   # keep them: work may be queued
                    # a priority pair that failed is replaced by an equal-priority, probed pair (the kind that has never been
                    # seen to starve a half); an equal-priority pair by another one
                    self._pair_mode = "probe"
                    self.streams = overlapping_streams(self.device, len(self.parts), mode=self._pair_mode)
                    self.stream_redraws += 1
                    self._serial_seen = 0
                    self._watch_left = 4
                    self._watch.clear()
                    return
            else:
                self._serial_seen = 0

    @property
    def graph_retry_off(self) -> bool:
        return any(p.graph_retry_off for p in self.parts)

    def get_timing(self, reset: bool = False) -> Dict[str, Dict[str, float]]:
        """Sum over the parts (their searches overlap on the device, so these are stream-busy times, not wall time)."""
        out: Dict[str, Dict[str, float]] = {"timing_ms": {}, "timing_calls": {}}
        for p in self.parts:
            t = p.get_timing(reset)
            for k in out:
                for name, v in t[k].items():
                    out[k][name] = out[k].get(name, 0) + v
        return out

    def search_batch(self, state: GpuStateBatch, *, temperatures: torch.Tensor, active: Optional[torch.Tensor] = None,
                     add_dirichlet_noise: Optional[bool] = None, reset: Optional[torch.Tensor] = None,
                     played_action: Optional[torch.Tensor] = None,
                     force_uniform_random_mask: Optional[torch.Tensor] = None,
                     rng_game_ids: Optional[torch.Tensor] = None,
                     rng_plies: Optional[torch.Tensor] = None, compact: Optional[bool] = None) -> RootSearchBatchOutput:
        main = torch.cuda.current_stream(self.device)
        cut = lambda t, a, b: None if t is None else t[a:b]
        outs = []
        self._check_overlap()
        # the reference search of a run: the 4th (the graphs of both launch keys exist by then), both halves on the caller's
        # stream; retried at the next search if a part still captured something in it
        calib = (not self.serialize and len(self.parts) == 2 and self._t_serial is None and self._searches >= 4 and
                 not any(len(w) > 4 and w[4] for w in self._watch))
        streams = (main,) * len(self.parts) if (self.serialize or calib) else self.streams
        subs = [state._map(lambda t, a=a, b=b: t[a:b]) for a, b in self.bounds]
        watch = None
        if not self.serialize and len(self.parts) == 2 and (self._watch_left > 0 or calib):
            watch = [torch.cuda.Event(enable_timing=True) for _ in range(4)]           # start / end of the two searches
        captures = sum(getattr(p, "captures", 0) for p in self.parts)
        t_host = time.perf_counter()
        for i, ((a, b), part, st, sub) in enumerate(zip(self.bounds, self.parts, streams, subs)):     # launch every part first ...
            st.wait_stream(main)
            with torch.cuda.stream(st):
                if watch is not None:
                    watch[2 * i].record(st)
                part.launch_search(sub, active=cut(active, a, b), add_dirichlet_noise=add_dirichlet_noise,
                                   reset=cut(reset, a, b), played_action=cut(played_action, a, b),
                                   rng_game_ids=cut(rng_game_ids, a, b), rng_plies=cut(rng_plies, a, b), compact=compact)
                if watch is not None:
                    watch[2 * i + 1].record(st)
        # A search in which a part captured its graph (warm-up, host synchronisation, capture) finishes part 0 on the host
        # before part 1 starts: its bracket says "serial" whatever the streams do -- with subtree reuse the first two
        # searches of every run capture (fresh key, continued key) and drew a spurious new pair at the third
        # ... and so does a search whose two launches the HOST spread out (another thread of the process held the runtime: the
        # streaming worker's copier pins staging memory, ~80 ms per GB): part 1 then starts late for a reason no other pair of
        # streams would cure.  Two graph replays take a few hundred microseconds to launch.
        host_ms = (time.perf_counter() - t_host) * 1e3
        if watch is not None and sum(getattr(p, "captures", 0) for p in self.parts) == captures and (host_ms < 6.0 or calib):
            self._watch.append(watch + [calib])
        todo = list(zip(self.parts, streams))                  # leftover rounds (batch_k > 1): the parts take turns,
        while todo:                                            # so that their small rounds overlap on the device
            nxt = []
            for part, st in todo:
                with torch.cuda.stream(st):
                    if part.tail_step():
                        nxt.append((part, st))
            todo = nxt
        for (a, b), part, st, sub in zip(self.bounds, self.parts, streams, subs):     # ... then the parts that may wait
            with torch.cuda.stream(st):
                outs.append(part.complete_search(sub, temperatures=temperatures[a:b],
                                                 force_uniform_random_mask=cut(force_uniform_random_mask, a, b),
                                                 tails_done=True))
        if not self.serialize:
            for st in self.streams:
                main.wait_stream(st)
        return RootSearchBatchOutput(*(torch.cat([getattr(o, f) for o in outs], dim=0)
                                       for f in ("model_input", "legal_mask", "policy_dense", "root_value", "terminal_mask",
                                                 "chosen_action_indices", "chosen_action_codes", "chosen_valid_mask")))


class SteadyStateTreeSelfPlay:
    """B games, always full (finished games are re-seated); one step = one searched move for every game."""

    def __init__(self, model, num_games: int, sims: int, device, dtype: str = "float16", seed: int = 12345,
                 temperature_init: float = 1.0, temperature_final: float = 0.1, temperature_threshold: int = 10,
                 max_game_plies: int = 512, exploration_weight: float = 1.0, reuse_tree: bool = False,
                 reuse_factor: float = -1.0, dual_stream: bool = False, batch_k: int = 1,
                 arena_rows: Optional[int] = None) -> None:
        """`arena_rows`: trajectory rows to preallocate (one per game and step; the arena doubles -- one host sync and a
        copy -- when it runs out, so timed runs size it for all their steps)."""
        from .steady_state import SteadyStateRootSelfPlay
        from .mcts_gpu import V1RootMCTSConfig
        dev = torch.device(device)
        self.net = model if isinstance(model, FusedNet) else FusedNet(model, dev)
        self.dev, self.B, self.sims = dev, int(num_games), int(sims)
        # reuse the population bookkeeping (states, plies, trajectory arena, preroll, reset)
        self.pop = SteadyStateRootSelfPlay(self.net, num_games, V1RootMCTSConfig(num_simulations=1), dev, seed=seed,
                                           temperature_init=temperature_init, temperature_final=temperature_final,
                                           temperature_threshold=temperature_threshold, max_game_plies=max_game_plies,
                                           fused_search=False, arena_rows=arena_rows)
        # two streams pay for 64 channels (two 4-wave workgroups share a CU); the 128-channel kernel fills a CU's registers
        # with one workgroup, so its launches cannot overlap -- LZ_DUAL_128=1 tries it anyway (experiment)
        allow = self.net.pack.channels == 64 or os.environ.get("LZ_DUAL_128", "0") == "1"
        self.dual_stream = bool(dual_stream and allow and int(num_games) >= 2
                                and not persistent_search_available(self.net, batch_k))
        if self.dual_stream:
            parts = int(os.environ.get("LZ_TREE_PARTS", "0")) or None           # experiment: more than two parts
            self.mcts = DualStreamTreeMCTS(self.net, num_games, sims, dev, num_parts=parts,
                                           exploration_weight=exploration_weight,
                                           reuse_tree=reuse_tree, reuse_factor=reuse_factor, batch_k=batch_k, seed=seed)
        else:
            one = self.net
            if os.environ.get("LZ_SINGLE_STREAM_HALF_WG", "0") == "1":      # experiment: 4-wave workgroups, two per CU, one stream
                one = self.net.variant(half_workgroups=True)
            self.mcts = PortableTreeMCTS(one, num_games, sims, dev, exploration_weight, reuse_tree=reuse_tree,
                                         reuse_factor=reuse_factor, batch_k=batch_k, seed=seed)
        self._reseated = torch.zeros((self.B,), dtype=torch.uint8, device=dev)
        self.positions = 0
        self._nn_events = []

    @property
    def leaf_evals(self) -> int:
        return self.mcts.leaf_evals

    @property
    def consumed_evals(self) -> int:
        return self.mcts.consumed_evals

    def preroll(self, n: int = 120) -> None:
        self.pop.preroll(n)

    def prepare(self) -> None:
        """Kernel loading and graph capture before anything is timed (the searches run here are discarded)."""
        self.mcts.prepare(self.pop.states)
        for e in ([self.mcts.engine] if hasattr(self.mcts, "engine") else [p.engine for p in self.mcts.parts]):
            e.reuse_dropped.zero_()
        torch.cuda.synchronize(self.dev)

    def step(self) -> None:
        p = self.pop
        temps = torch.where(p.plies < p.t_thr, p.t_init, p.t_final).to(torch.float32)
        search = self.mcts.search_batch(p.states, temperatures=temps, reset=self._reseated, rng_plies=p.plies)
        self._reseated.zero_()
        p.finish_step(search, reseated=self._reseated)
        self.positions += self.B


from collections import OrderedDict

_ENGINE_CACHE: "OrderedDict[tuple, object]" = OrderedDict()


def _engine_cache_limit() -> int:
    """Search engines kept alive between self_play_tree_gpu calls (LZ_ENGINE_CACHE, default 1; 0 = off).  An engine owns
    its arenas (C3: ~90 GB), so the default keeps only the last one."""
    try:
        return max(0, int(os.environ.get("LZ_ENGINE_CACHE", "1")))
    except ValueError:
        return 1


def clear_engine_cache() -> None:
    """Drop the cached engines (and give their arenas back to the allocator)."""
    _ENGINE_CACHE.clear()
    torch.cuda.empty_cache()


def self_play_tree_gpu(model, num_games: int, mcts_simulations: int, temperature_init: float, temperature_final: float,
                       temperature_threshold: int, exploration_weight: float, device: str,
                       add_dirichlet_noise: bool = True, dirichlet_alpha: float = 0.3, dirichlet_epsilon: float = 0.25,
                       soft_value_k: float = 2.0, opening_random_moves: int = 0, max_game_plies: int = 512,
                       sample_moves: bool = True, concurrent_games: int = 8, verbose: bool = False,
                       policy_target_temperature: Optional[float] = None,
                       policy_target_prior_pseudocount: float = 0.0, reuse_tree: bool = True,
                       reuse_factor: float = -1.0, dual_stream: Optional[bool] = None, continuous_waves: bool = True,
                       device_tail: bool = True, batch_k: int = 1, evaluator: str = "auto", seed: int = 12345,
                       collect_timing: bool = False, row_log=None) -> Tuple[TensorSelfPlayBatch, SelfPlayV1Stats]:
    """Tree-search twin of self_play_v1_gpu (same outputs); mirrors v1/python/portable_self_play.py:82-284,
    including the subtree reuse it performs on every move (:191, `reuse_tree`).
    `model` may also be a `PriorEvaluator` (states -> priors over the 220 actions + values, the reference's own
    split-phase hand-off).  `evaluator`: "fused" = the hand-written fp16 network kernel inside the captured search (6x64 / 10x128 nets);
    "module" = `model` itself as an external fp32 evaluator behind the split-phase protocol, whatever its size or
    device (what the reference's portable runner does with its `model`); "auto" = fused when the net has a fused
    kernel.  `seed`: key of the per-game counter RNG (noise, sampled moves); game ids are the runner's game numbers.
    `row_log` (finished_log.FinishedRowLog, the streaming worker): the rows of every game leave through the log when the
    game ends and the returned batch is empty; needs `device_tail` and `continuous_waves`."""
    dev = torch.device(device)
    if row_log is not None and not (device_tail and continuous_waves):
        raise RuntimeError("self_play_tree_gpu: a finished-row log needs device_tail and continuous_waves")
    if evaluator not in ("auto", "fused", "module"):
        raise ValueError(f"evaluator must be auto / fused / module, got {evaluator!r}")
    if isinstance(model, FusedNet):
        net, use_fused = model, True
    elif isinstance(model, PriorEvaluator):
        net, use_fused = model, False
    else:
        from .net_hip import fused_supported
        use_fused = evaluator == "fused" or (evaluator == "auto" and fused_supported(model))
        net = FusedNet(model, dev) if use_fused else model.eval()
    wave = max(1, min(int(concurrent_games), int(num_games)))
    # two half-batches on two streams (see DualStreamTreeMCTS) once a wave is large enough to fill the chip twice over
    if dual_stream is None:
        dual_stream = use_fused and net.pack.channels == 64 and wave >= 1024 and int(batch_k) <= 1   # waves: large launches already
    if use_fused and persistent_search_available(net, batch_k):
        dual_stream = False                                  # the persistent kernel overlaps the phases inside every CU
    cls = DualStreamTreeMCTS if (dual_stream and use_fused and net.pack.channels == 64 and wave >= 2) else PortableTreeMCTS
    t_setup = time.perf_counter()
    kw = dict(exploration_weight=float(exploration_weight), add_dirichlet_noise=bool(add_dirichlet_noise),
              dirichlet_alpha=float(dirichlet_alpha), dirichlet_epsilon=float(dirichlet_epsilon),
              sample_moves=bool(sample_moves), reuse_tree=bool(reuse_tree), reuse_factor=float(reuse_factor),
              policy_target_temperature=policy_target_temperature,
              policy_target_prior_pseudocount=float(policy_target_prior_pseudocount), batch_k=int(batch_k),
              collect_timing=bool(collect_timing))
    # Engines (arenas, descriptors) and their captured graphs are kept between calls with the same network buffers and
    # shape -- the worker calls this once per chunk and the staged loop once per iteration, and construction + capture
    # were 12.5 % of a C2 run.  `FusedNet.refresh()` writes new weights into the same buffers, so a
    # cached graph stays valid across checkpoints.  A PriorEvaluator / external module is never cached.
    # Only a caller-owned FusedNet is a stable key: for a module a fresh FusedNet is packed per call, its engine could
    # never be found again and would only pin its arenas.  Engines stay alive after the return -- call
    # `clear_engine_cache()` before handing the GPU's memory to something else (training on the same device).
    key = None
    if use_fused and isinstance(model, FusedNet) and _engine_cache_limit() > 0:
        key = (cls.__name__, int(net.pack.wfrag.data_ptr()), int(net.pack.fparams.data_ptr()), int(net.desc.flags),
               str(dev), wave, int(mcts_simulations), tuple(sorted((k, repr(v)) for k, v in kw.items())),
               os.environ.get("LZ_TREE_GRAPH", "on"), os.environ.get("LZ_TREE_PERSISTENT", "0"))
    mcts = _ENGINE_CACHE.pop(key, None) if key is not None else None
    cache_hit = mcts is not None
    if mcts is None:
        if key is not None:
            while len(_ENGINE_CACHE) >= _engine_cache_limit():      # make room BEFORE allocating the new arenas
                _ENGINE_CACHE.popitem(last=False)
            torch.cuda.empty_cache()
        mcts = cls(net, wave, mcts_simulations, dev, seed=seed, **kw)
    else:
        mcts.reset_run(seed)
    if key is not None:
        _ENGINE_CACHE[key] = mcts
    setup_sec = time.perf_counter() - t_setup
    buffer = TensorTrajectoryBuffer(dev, TOTAL_ACTION_DIM, max_steps_hint=max_game_plies, concurrent_games_hint=wave)
    outcome = torch.zeros((3,), dtype=torch.int64, device=dev)
    lengths = torch.zeros((int(num_games),), dtype=torch.int64, device=dev)
    # device_tail: rows / move / finalisation on the device (wave_tail.WaveTail); continuous_waves: one wave whose
    # finished slots start the remaining games at once instead of sequential waves that idle until their longest game
    tail = None
    delta_hist = None
    if device_tail:
        from .wave_tail import WaveTail
        tail = WaveTail(buffer, wave, int(max_game_plies), dev, soft_value_k=float(soft_value_k), row_log=row_log)
        tail.collect_timing = bool(collect_timing)
        outcome, delta_hist = tail.outcome, tail.delta_hist
    continuous = tail is not None and bool(continuous_waves)
    wasted_plies = 0
    started = time.perf_counter()
    for base in range(0, wave if continuous else int(num_games), wave):
        g = min(wave, int(num_games) - base)
        states = GpuStateBatch.initial(dev, wave)
        step_index = None if row_log is not None else torch.full((wave, max_game_plies), -1, dtype=torch.int64, device=dev)
        step_counts = torch.zeros((wave,), dtype=torch.int64, device=dev)
        plies = torch.zeros((wave,), dtype=torch.int64, device=dev)
        done = torch.zeros((wave,), dtype=torch.bool, device=dev)
        if g < wave:
            done[g:] = True
        mcts.reset_trees()
        slot_ids = torch.arange(wave, dtype=torch.int64, device=dev)
        if tail is not None:
            force_n = int(opening_random_moves)
            # compact evaluation lists once the wave has drained by at least one network pass per CU (the engine was built
            # for both launch forms if that can happen at all; two plies of lag in the estimate only delay the switch)
            lists_below = wave - (samples_per_launch_pass(net) if use_fused else 0)
            tail.run(lambda st, temps, dn, reseated: mcts.search_batch(
                         st, temperatures=temps, active=~dn, reset=reseated,
                         force_uniform_random_mask=(plies < force_n) if force_n > 0 else None,
                         rng_game_ids=tail.slot_game + (0 if continuous else base), rng_plies=plies,
                         **({"compact": True} if (use_fused and tail.live_estimate <= lists_below) else {})),
                     states, plies, done, step_index, step_counts, lengths if continuous else lengths[base:base + g],
                     temperature_init, temperature_final, temperature_threshold,
                     games_to_start=int(num_games) - wave if continuous else 0)
            wasted_plies += 1
            continue
        while True:
            active = torch.nonzero(~done).view(-1)
            if int(active.numel()) == 0:
                break
            temps = torch.where(plies < int(temperature_threshold), float(temperature_init),
                                float(temperature_final)).to(torch.float32)
            force = (plies < int(opening_random_moves)) if int(opening_random_moves) > 0 else None
            out = mcts.search_batch(states, temperatures=temps, active=~done, force_uniform_random_mask=force,
                                    rng_game_ids=slot_ids + base, rng_plies=plies)
            rows = buffer.append_steps(out.model_input.index_select(0, active), out.legal_mask.index_select(0, active),
                                       out.policy_dense.index_select(0, active),
                                       states.current_player.index_select(0, active))
            step_index[active, step_counts.index_select(0, active)] = rows
            step_counts.index_add_(0, active, torch.ones_like(active))
            fin, result, _ = v0_core.self_play_step_inplace(
                *states.tensors(), plies, done, active, out.chosen_action_codes.index_select(0, active),
                out.terminal_mask.index_select(0, active), out.chosen_valid_mask.index_select(0, active),
                int(max_game_plies), float(soft_value_k))
            if int(fin.numel()) > 0:
                boards = states.board.index_select(0, fin)
                delta = (boards.eq(1).sum(dim=(1, 2)) - boards.eq(-1).sum(dim=(1, 2))).to(torch.float32)
                soft = torch.tanh(delta / 18.0 * float(soft_value_k))
                f_slots, f_len, f_out = buffer.finalize_games_inplace(
                    step_index_matrix=step_index, step_counts=step_counts, slots=fin, result_from_black=result,
                    soft_value_from_black=soft)
                if int(f_slots.numel()) > 0:
                    lengths.index_copy_(0, f_slots + base, f_len)
                outcome.add_(f_out)
    t_sync = time.perf_counter()
    torch.cuda.synchronize(dev)
    sync_ms = (time.perf_counter() - t_sync) * 1e3
    elapsed = max(1e-9, time.perf_counter() - started)
    if tail is not None:
        tail.check_overflow()
    if row_log is not None:
        row_log.close()                                      # the last segment leaves; the rows are the consumer's now
    t_build = time.perf_counter()
    batch = buffer.build()
    torch.cuda.synchronize(dev)
    build_sec = time.perf_counter() - t_build
    positions = int(lengths.sum().item()) if row_log is not None else batch.num_samples
    o = outcome.tolist()
    hist = delta_hist.tolist() if delta_hist is not None else []
    keys = ("root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms")
    timing = mcts.get_timing() if collect_timing else {"timing_ms": {}, "timing_calls": {}}
    if tail is not None and collect_timing:
        for k, v in tail.get_timing().items():
            timing["timing_ms"][k] = timing["timing_ms"].get(k, 0.0) + v["ms"]
            timing["timing_calls"][k] = timing["timing_calls"].get(k, 0) + v["calls"]
    t_ms = {k: float(timing["timing_ms"].get(k, 0.0)) for k in keys}
    t_total = sum(t_ms.values())
    # what the bounded arenas did to this run (the reference's tree is unbounded): must all be 0 for results that equal an
    # unbounded tree's; a refused expansion means the edge pool was sized too small for this workload -- say so loudly
    engines = [p.engine for p in getattr(mcts, "parts", [])] or [mcts.engine]
    compact_lists = all(e.compact_evals for e in engines)
    refused = sum(e.pool_status()["refused_expansions"] for e in engines)
    dropped = sum(int(e.reuse_dropped[0].item()) for e in engines)
    pruned = sum(int(e.reuse_dropped[1].item()) for e in engines)
    if refused:
        print(f"[liuzhou_amd] WARNING: {refused} tree expansions found no free edge chunk (pool of "
              f"{sum(e.pool_chunks for e in engines)} chunks); those leaves stayed unexpanded for a visit.  Pass a larger "
              "`pool_chunks` to TreeEngine or lower `concurrent_games`.", flush=True)
    stats = SelfPlayV1Stats(
        num_games=num_games, num_positions=positions, black_wins=int(o[0]), white_wins=int(o[1]), draws=int(o[2]),
        avg_game_length=float(lengths.to(torch.float32).mean().item()), elapsed_sec=elapsed,
        positions_per_sec=float(positions / elapsed), games_per_sec=float(num_games / elapsed),
        step_timing_ms=t_ms, step_timing_ratio={k: (t_ms[k] / t_total if t_total > 0 else 0.0) for k in keys},
        step_timing_calls={k: int(timing["timing_calls"].get(k, 0)) for k in keys},
        # the device-tail loop runs one fully masked ply per wave after the last game has ended (wave_tail.WaveTail.run)
        # (with compact evaluation lists a fully masked ply launches no evaluation at all)
        # (... unless that ply was launched densely: the two-plies-old live estimate had not seen the wave end yet)
        mcts_counters={"leaf_eval_count": int(mcts.leaf_evals) - (
                           wasted_plies * wave * (int(mcts_simulations) + 1)
                           if int(batch_k) <= 1 and not (compact_lists and getattr(mcts, "last_search_lists", False)) else 0),
                       "compact_eval_lists": int(compact_lists), "search_parts": len(engines),
                       "list_searches": int(sum(getattr(p, "list_searches", 0) for p in (getattr(mcts, "parts", None) or [mcts]))),
                       "masked_extra_plies": wasted_plies, "graph_retry_off": int(bool(mcts.graph_retry_off)),
                       # where the wall time outside `elapsed_sec` (the plies) goes: engine construction or cache hit,
                       # build() of the five tensors; graph capture happens inside the first plies
                       "engine_cache_hit": int(cache_hit), "setup_ms": int(setup_sec * 1e3), "build_ms": int(build_sec * 1e3),
                       "reuse_pruned": pruned, "reuse_dropped": dropped, "edge_pool_refused": refused,
                       "final_sync_ms": int(sync_ms), "stream_redraws": int(getattr(mcts, "stream_redraws", 0)),
                       # two-stream search: last watched search / the run's one-stream reference search, in percent
                       # (~66: the halves share the chip; ~100: they do not)
                       **({"stream_overlap_pct": int(round(100.0 * mcts.overlap_ratio))}
                          if getattr(mcts, "overlap_ratio", None) is not None else {}),
                       **({"loop_ms": int(tail.loop_ms), "host_wait_ms": int(tail.host_wait_ms),
                           "plies_launched": int(tail.plies_launched)} if tail is not None else {})},
        piece_delta_buckets={str(d - 18): int(v) for d, v in enumerate(hist)}, device=str(dev))
    return batch, stats
