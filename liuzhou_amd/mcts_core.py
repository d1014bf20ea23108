"""`MCTSConfig` / `MCTSCore` / `InferenceEngine` / `EvalBatcher` / `TorchScriptRunner`: the class surface of the reference's `v0_core` that `v0/python/mcts.py`
(:138-470) and the web backend's model loader bind, as thin adapters over the device-resident tree engine (SURVEY.md
section 8 row f4).

  reference                                                    here
  v0/src/bindings/module.cpp:1158-1173  MCTSConfig              `MCTSConfig` (same fields / defaults)
  v0/src/bindings/module.cpp:1175-1284  MCTSCore                `MCTSCore`: one game on a `TreeEngine(1, ...)`
  v0/src/bindings/module.cpp:1422-1438  InferenceEngine         `InferenceEngine`: the fused network kernel (or any module)
  v0/src/bindings/module.cpp:1440-1469  EvalBatcher             `EvalBatcher`: worker thread packing callers' inputs into the engine's batch
  v0/src/bindings/module.cpp:1471-1479  TorchScriptRunner       `TorchScriptRunner`: the archive as is on the host, the fused kernel on a HIP device
  v0/src/mcts/mcts_core.cpp:181-230,703-760,815-829             root_value / get_policy / get_root_children_stats / advance_root

Search semantics are the engine's variant P (one leaf per simulation, first maximum wins, sign flipped only when the mover
changes) -- the reference's own C++ tree flips the sign on every level, which is wrong for this game's same-player atomic
phases ("not an oracle"), and uses virtual-loss batches; `batch_size` / `virtual_loss` /
`max_actions_per_batch` are accepted and ignored.  TorchScript is not loaded: `InferenceEngine` takes a module, a
state-dict file or a TorchScript archive whose state_dict has ChessNet's keys, and rebuilds the network from it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from .mcts_gpu import GpuStateBatch
from .net import ChessNet
from .net_hip import FusedNet
from .tree_engine import MAX_CHILDREN, OUT_CAP, TreeEngine


@dataclass
class MCTSConfig:
    """v0/include/v0/mcts_core.hpp:18-32 (defaults as the PyBind class exposes them)."""
    num_simulations: int = 800
    exploration_weight: float = 1.0
    temperature: float = 1.0
    add_dirichlet_noise: bool = False
    dirichlet_alpha: float = 0.3
    dirichlet_epsilon: float = 0.25
    batch_size: int = 16
    max_actions_per_batch: int = 0
    virtual_loss: float = 1.0
    seed: int = 12345
    device: str = "cuda"


def _int(v) -> int:
    return int(getattr(v, "value", v))


def state_like_to_batch(state, device) -> GpuStateBatch:
    """module.cpp:79-178 (GameStateFromPyLike / CoerceGameStateLike): any object with `board` (6x6 nested ints),
    `phase`, `current_player` (ints or enums), optional `marked_black` / `marked_white` (iterables of (r, c)) and the
    optional counters.  Like the reference's coercion, `moves_since_capture` is NOT read (it stays 0)."""
    if state is None:
        raise RuntimeError("state must not be None")
    for name in ("board", "phase", "current_player"):
        if not hasattr(state, name):
            raise RuntimeError(f"Python GameState is missing attribute: {name}")
    board = torch.tensor([[int(v) for v in row] for row in state.board], dtype=torch.int8)
    if tuple(board.shape) != (6, 6):
        raise RuntimeError(f"board must be 6x6, got {tuple(board.shape)}")
    phase, player = _int(state.phase), _int(state.current_player)
    if not 1 <= phase <= 7:
        raise RuntimeError("phase enum value out of range")
    if player not in (1, -1):
        raise RuntimeError("current_player must be 1 or -1")

    def marks(name):
        m = torch.zeros((6, 6), dtype=torch.bool)
        for rc in (getattr(state, name, None) or ()):
            r, c = int(rc[0]), int(rc[1])
            if not (0 <= r < 6 and 0 <= c < 6):
                raise RuntimeError("mark coordinate outside the board")
            m[r, c] = True
        return m

    opt = lambda name: int(getattr(state, name, 0) or 0)
    one = lambda v: torch.tensor([int(v)], dtype=torch.int64)
    b = GpuStateBatch(board.view(1, 6, 6), marks("marked_black").view(1, 6, 6), marks("marked_white").view(1, 6, 6),
                      one(phase), one(player), one(opt("pending_marks_required")), one(opt("pending_marks_remaining")),
                      one(opt("pending_captures_required")), one(opt("pending_captures_remaining")),
                      one(opt("forced_removals_done")), one(opt("move_count")), one(0))
    return b.to(device)


class InferenceEngine:
    """module.cpp:1422-1438: `InferenceEngine(path, device, dtype, batch_size, ...)` with `.forward(input, n_valid)`.
    `path` may be a ChessNet module, a state-dict file, or a TorchScript archive (only its state_dict is used: the
    reference's TorchScript / CUDA-graph replay, v0/src/net/inference_engine.cpp:58-202, is replaced by the fused network
    kernel for 64- / 128-channel nets and by the eager module otherwise)."""

    def __init__(self, path, device: str = "cuda", dtype: str = "float16", batch_size: int = 512,
                 input_channels: int = 11, height: int = 6, width: int = 6, warmup_iters: int = 5,
                 use_inference_mode: bool = True) -> None:
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise RuntimeError("InferenceEngine needs a HIP device (no CPU path)")
        self._dtype, self._batch = str(dtype), int(batch_size)
        model = path if isinstance(path, torch.nn.Module) else self._load(path)
        self.model = model.to(self._device).eval()
        from .net_hip import fused_supported
        self.fused: Optional[FusedNet] = None
        if fused_supported(self.model) and self._dtype in ("float16", "half", "fp16", "float32", "fp32"):
            self.fused = FusedNet(self.model, self._device,
                                  precision="fp32" if self._dtype in ("float32", "fp32") else "fp16")

    @staticmethod
    def _load(path) -> torch.nn.Module:
        try:
            sd = torch.jit.load(str(path), map_location="cpu").state_dict()
        except Exception:
            obj = torch.load(str(path), map_location="cpu", weights_only=False)
            sd = obj.get("model_state_dict", obj.get("state_dict", obj)) if isinstance(obj, dict) else obj.state_dict()
        c = int(sd["stem_conv.weight"].shape[0])
        nb = len({k.split(".")[1] for k in sd if k.startswith("blocks.")})
        model = ChessNet(trunk_channels=c, num_blocks=nb,
                         policy_channels=int(sd["policy_head.conv1.weight"].shape[0]),
                         value_channels=int(sd["value_head.conv1.weight"].shape[0]),
                         value_mlp_channels=int(sd["value_head.fc1.weight"].shape[0]),
                         value_bucket_bins=int(sd["value_head.fc2.weight"].shape[0]))
        model.load_state_dict(sd, strict=True)
        return model

    def forward(self, input: torch.Tensor, n_valid: int = -1):
        """-> (log_p1, log_p2, log_pmc, value output of the network) for the first n_valid rows (all: -1)."""
        x = input.to(self._device, torch.float32)
        if n_valid is not None and int(n_valid) >= 0:
            x = x[: int(n_valid)]
        if self.fused is not None:
            return self.fused(x)
        with torch.inference_mode():
            return self.model(x)

    device = property(lambda self: str(self._device))
    dtype = property(lambda self: self._dtype)
    batch_size = property(lambda self: self._batch)
    graph_enabled = property(lambda self: self.fused is not None)


_DTYPE_NAMES = {"float32": torch.float32, "fp32": torch.float32, "f32": torch.float32, "float16": torch.float16,
                "fp16": torch.float16, "f16": torch.float16, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


class TorchScriptRunner:
    """module.cpp:1471-1479 over v0/src/net/torchscript_runner.cpp:55-109: `TorchScriptRunner(path, device="cpu",
    dtype="auto", use_inference_mode=True)`, `.forward(input)` -> (log_p1, log_p2, log_pmc, value output), `.device`,
    `.dtype`.  On a host device the archive runs as the reference runs it (`torch.jit.load`, the module moved to `dtype`, the
    input converted to it or left in its own dtype for "auto").  On a HIP device the archive is not interpreted: its
    state_dict must have ChessNet's keys, the network is rebuilt from it and evaluated by the fused network kernel
    (64 / 128 channels; fp16 operands for "auto" / "float16", fp32 operands for "float32") -- any other archive raises."""

    def __init__(self, path, device: str = "cpu", dtype: str = "auto", use_inference_mode: bool = True) -> None:
        self._device = torch.device(device)
        name = str(dtype)
        if name in ("", "auto", "none"):
            self._dtype: Optional[torch.dtype] = None
        elif name in _DTYPE_NAMES:
            self._dtype = _DTYPE_NAMES[name]
        else:
            raise RuntimeError("Unsupported dtype: " + name)
        if self._device.type == "cpu" and self._dtype == torch.float16:
            raise RuntimeError("float16 is not supported on CPU for TorchScriptRunner.")
        self._inference = bool(use_inference_mode)
        self.fused: Optional[FusedNet] = None
        self.module = None
        if self._device.type == "cpu":
            self.module = torch.jit.load(str(path), map_location=self._device)
            if self._dtype is not None:
                self.module.to(self._device, self._dtype)
            self.module.eval()
        else:
            if self._dtype == torch.bfloat16:
                raise RuntimeError("TorchScriptRunner: the fused network kernel computes in float16 or float32")
            model = InferenceEngine._load(path).to(self._device).eval()
            from .net_hip import fused_unsupported_reason
            why = fused_unsupported_reason(model)
            if why is not None:
                raise RuntimeError(f"TorchScriptRunner: the fused network kernel does not cover this model: {why}")
            self.fused = FusedNet(model, self._device, precision="fp32" if self._dtype == torch.float32 else "fp16")

    def forward(self, input: torch.Tensor):
        if self.fused is not None:
            return self.fused(input.to(self._device, torch.float32))
        x = input
        want = self._dtype if self._dtype is not None else x.dtype
        if x.device != self._device or x.dtype != want:
            x = x.to(self._device, want)
        x = x.contiguous()
        ctx = torch.inference_mode() if self._inference else torch.no_grad()
        with ctx:
            out = self.module(x)
        if not isinstance(out, (tuple, list)) or len(out) != 4:
            raise RuntimeError("TorchScriptRunner expected 4 outputs (log_p1, log_p2, log_pmc, value).")
        return tuple(out)

    device = property(lambda self: str(self._device))
    dtype = property(lambda self: "auto" if self._dtype is None else {torch.float32: "float32", torch.float16: "float16",
                                                                      torch.bfloat16: "bfloat16"}[self._dtype])


class EvalBatcher:
    """module.cpp:1440-1469 over v0/src/mcts/eval_batcher.cpp:27-272: callers on several threads hand in small inputs,
    one worker thread packs what is waiting into the engine's fixed batch (first come first packed, a request that does not
    fit starts the next batch, `timeout_ms` after the first request the batch goes as it is), runs ONE `engine.forward(buffer,
    n_valid)` and hands every caller its rows.  Same constructor checks, per-request errors, statistics
    (`eval_calls`, `eval_leaves`, `full512_calls`, 17-bucket histogram) and shutdown behaviour; `engine` is an
    `InferenceEngine` (anything with `forward(input, n_valid)`, `batch_size` and `device`)."""

    HIST_BUCKETS = 17

    def __init__(self, engine, batch_size: int = 512, input_channels: int = 11, height: int = 6, width: int = 6,
                 timeout_ms: int = 2) -> None:
        import collections
        import threading
        if engine is None:
            raise RuntimeError("EvalBatcher requires a valid InferenceEngine.")
        self._engine, self._batch, self._timeout = engine, int(batch_size), int(timeout_ms)
        self._shape = (int(input_channels), int(height), int(width))
        if self._batch <= 0:
            raise RuntimeError("EvalBatcher batch_size must be positive.")
        if min(self._shape) <= 0:
            raise RuntimeError("EvalBatcher input shape must be positive.")
        if int(engine.batch_size) != self._batch:
            raise RuntimeError(f"EvalBatcher batch_size mismatch: engine={int(engine.batch_size)} batcher={self._batch}")
        self._device = torch.device(engine.device)
        self._buf: Optional[torch.Tensor] = None
        self._queue = collections.deque()
        self._cv = threading.Condition()
        self._stop = False
        self._stats_lock = threading.Lock()
        self._zero_stats()
        self._worker = threading.Thread(target=self._loop, name="lz-eval-batcher", daemon=True)
        self._worker.start()

    def _zero_stats(self) -> None:
        self._calls = self._leaves = self._full = 0
        self._hist = [0] * self.HIST_BUCKETS

    def forward(self, input: torch.Tensor, n_valid: int = -1):
        from concurrent.futures import Future
        if self._stop:
            raise RuntimeError("EvalBatcher is shut down.")
        fut: Future = Future()
        with self._cv:
            self._queue.append([input, int(n_valid), fut])
            self._cv.notify()
        return fut.result()

    def shutdown(self) -> None:
        with self._cv:
            if self._stop:
                return
            self._stop = True
            self._cv.notify_all()
        self._worker.join()

    def __del__(self) -> None:
        try:
            self.shutdown()
        except Exception:
            pass

    def reset_eval_stats(self) -> None:
        with self._stats_lock:
            self._zero_stats()

    def get_eval_stats(self) -> Dict[str, object]:
        with self._stats_lock:
            return {"eval_calls": self._calls, "eval_leaves": self._leaves, "full512_calls": self._full,
                    "hist": list(self._hist)}

    batch_size = property(lambda self: self._batch)
    timeout_ms = property(lambda self: self._timeout)

    def _bucket(self, n: int) -> int:                                # eval_batcher.cpp:112-127
        if n <= 0:
            return 0
        if n > self._batch:
            return self.HIST_BUCKETS - 1
        return min(self.HIST_BUCKETS - 1, max(0, (n - 1) // max(1, self._batch // (self.HIST_BUCKETS - 1))))

    def _admit(self, req) -> Optional[str]:
        """Validate one request (eval_batcher.cpp:163-196); returns the error text, or None with req[1] = its row count."""
        x, n = req[0], req[1]
        if n <= 0:
            n = int(x.shape[0]) if x.dim() > 0 else 0
        req[1] = n
        if n <= 0 or n > self._batch:
            return "EvalBatcher n_valid out of range."
        if x.dim() != 4:
            return "EvalBatcher input must be 4D."
        if tuple(int(v) for v in x.shape[1:]) != self._shape:
            return (f"EvalBatcher input shape mismatch, expected (B, {self._shape[0]}, {self._shape[1]}, {self._shape[2]}) "
                    f"got {tuple(int(v) for v in x.shape)}")
        if int(x.shape[0]) < n:
            return "EvalBatcher input batch smaller than n_valid."
        return None

    def _loop(self) -> None:
        import time
        while True:
            batch, total = [], 0
            with self._cv:
                self._cv.wait_for(lambda: self._stop or self._queue)
                if self._stop and not self._queue:
                    return
                deadline = time.monotonic() + max(0, self._timeout) / 1e3
                while total < self._batch:
                    limit = False
                    while self._queue and total < self._batch:
                        req = self._queue.popleft()
                        err = self._admit(req)
                        if err is not None:
                            req[2].set_exception(RuntimeError(err))
                            continue
                        if total + req[1] > self._batch:
                            self._queue.appendleft(req)
                            limit = True
                            break
                        batch.append(req)
                        total += req[1]
                    if total >= self._batch or limit or self._timeout <= 0:
                        break
                    if not self._queue:
                        left = deadline - time.monotonic()
                        if left > 0 and self._cv.wait_for(lambda: self._stop or self._queue, timeout=left):
                            if self._stop and not self._queue:
                                break
                            continue
                        break
            if not batch:
                if self._stop:
                    return
                continue
            self._run(batch, total)

    def _run(self, batch, total: int) -> None:
        if self._buf is None:
            self._buf = torch.zeros((self._batch,) + self._shape, dtype=torch.float32, device=self._device)
        self._buf.zero_()
        failed, off = [False] * len(batch), 0
        for i, (x, n, fut) in enumerate(batch):
            try:
                self._buf[off:off + n].copy_(x[:n].to(self._device, torch.float32), non_blocking=True)
            except Exception as exc:
                fut.set_exception(exc)
                failed[i] = True
            off += n
        try:
            outs = self._engine.forward(self._buf, total)
            with self._stats_lock:
                self._calls += 1
                self._leaves += total
                self._full += 1 if total == self._batch else 0
                self._hist[self._bucket(total)] += 1
            off = 0
            for i, (x, n, fut) in enumerate(batch):
                if not failed[i]:
                    fut.set_result(tuple(o[off:off + n].clone() for o in outs))
                off += n
        except Exception as exc:
            for i, (_x, _n, fut) in enumerate(batch):
                if not failed[i]:
                    fut.set_exception(exc)


class MCTSCore:
    """One search tree on the device engine behind the reference's `MCTSCore` methods."""

    def __init__(self, config: Optional[MCTSConfig] = None) -> None:
        self.cfg = config or MCTSConfig()
        self.device = torch.device(self.cfg.device)
        if self.device.type != "cuda":
            raise RuntimeError("MCTSCore needs a HIP device (no CPU path)")
        self._callback: Optional[Callable] = None
        self._fused: Optional[FusedNet] = None
        self._engine: Optional[TreeEngine] = None
        self._root: Optional[GpuStateBatch] = None
        self._root_like = None
        self._expanded = False
        self._sims_in_tree = 0
        self._eval_calls = self._eval_leaves = 0
        from .game_rng import GameRng
        self._rng = GameRng(1, self.device, seed=int(self.cfg.seed))
        self._noise = torch.zeros((1, OUT_CAP), dtype=torch.float32, device=self.device)

    # ---- evaluators (module.cpp:1177-1228) ----
    def set_forward_callback(self, callback: Callable) -> None:
        """callback(inputs f32[N,11,6,6]) -> (log_p1, log_p2, log_pmc, value f32[N])"""
        self._callback, self._fused = callback, None

    def set_inference_engine(self, engine: InferenceEngine) -> None:
        if engine is None:
            raise RuntimeError("InferenceEngine is null")
        if engine.fused is not None:
            self._fused, self._callback = engine.fused, None
        else:
            from .net import bucket_logits_to_scalar

            def cb(x, _e=engine):
                lp1, lp2, lpm, raw = _e.forward(x)
                return lp1, lp2, lpm, bucket_logits_to_scalar(raw.float())
            self._callback, self._fused = cb, None

    def _value_callback(self, forward: Callable) -> Callable:
        """A `forward(x) -> (log_p1, log_p2, log_pmc, value output)` as the search's callback (scalar values)."""
        from .net import bucket_logits_to_scalar

        def cb(x):
            lp1, lp2, lpm, raw = forward(x)
            raw = raw.float()
            return lp1.float(), lp2.float(), lpm.float(), (bucket_logits_to_scalar(raw) if raw.dim() == 2 and raw.shape[1] > 1 else raw.reshape(-1))
        return cb

    def set_torchscript_runner(self, runner) -> None:
        """module.cpp:1196-1217: evaluations go through `runner.forward` (the fused kernel when the runner sits on a HIP device)"""
        if runner is None:
            raise RuntimeError("TorchScriptRunner is null")
        if getattr(runner, "fused", None) is not None:
            self._fused, self._callback = runner.fused, None
        else:
            self._callback, self._fused = self._value_callback(lambda x, _r=runner: _r.forward(x)), None

    def set_eval_batcher(self, batcher) -> None:
        """module.cpp:1218-1230: evaluations go through `batcher.forward` -- several cores on several threads share one engine"""
        if batcher is None:
            raise RuntimeError("EvalBatcher is null")
        self._callback, self._fused = self._value_callback(lambda x, _b=batcher: _b.forward(x, int(x.shape[0]))), None

    # ---- tree ----
    NODE_LIMIT = 524288         # tree_advance_kernel marks a game's nodes in LDS: 8 192 words of 64 (csrc/lz_engine.hip; 65 536 until round 5)

    def _capacity(self) -> int:
        return max(1024, 4 * int(self.cfg.num_simulations))

    def set_root_state(self, state) -> None:
        self._root = state if isinstance(state, GpuStateBatch) else state_like_to_batch(state, self.device)
        self._root_like = state
        cap = self._capacity()
        if cap + 2 > self.NODE_LIMIT:
            # the reference's tree is unbounded; ours is an arena of at most 524 288 nodes per game
            raise ValueError(f"MCTSCore: num_simulations={self.cfg.num_simulations} needs an arena of {cap} nodes, above the "
                             f"{self.NODE_LIMIT}-node limit of a device tree (use num_simulations <= {(self.NODE_LIMIT - 2) // 4})")
        if self._engine is None or self._engine.max_sims < cap:
            # room for kept subtrees: up to 3 arenas' worth, never more than the node limit allows, never negative
            # (a negative factor would mean "size from free memory" to TreeEngine)
            factor = max(0.0, min(3.0, (self.NODE_LIMIT - cap - 2) / cap))
            self._engine = TreeEngine(1, cap, self.device, float(self.cfg.exploration_weight), reuse_factor=factor)
        self._engine.set_roots(self._root)
        self._engine.begin()
        self._expanded, self._sims_in_tree = False, 0
        # Noise key: the reference's generator is stateful (mcts_core.cpp:132,316), so every root expansion -- each
        # set_root_state, each advance_root -- draws fresh noise.  The key here is (seed, game 0, expansion counter): the
        # counter only ever grows, so two roots never share their Gamma draws.
        self._rng.ply.add_(1)

    def reset(self) -> None:
        self._root = self._root_like = None
        self._expanded, self._sims_in_tree = False, 0

    def _evaluate(self):
        e = self._engine
        if self._fused is not None:
            lp1, lp2, lpm, _, val = self._fused.forward_packed(e.buf["leaf_state"])
        else:
            if self._callback is None:
                raise RuntimeError("no forward callback / inference engine set")
            lp1, lp2, lpm, val = self._callback(e.leaf_planes())
            f = lambda t: t.float().reshape(1, -1).to(self.device).contiguous()
            lp1, lp2, lpm, val = f(lp1), f(lp2), f(lpm), val.float().reshape(-1).to(self.device).contiguous()
        self._eval_calls += 1
        self._eval_leaves += 1
        return lp1, lp2, lpm, val

    def run_simulations(self, num_simulations: int) -> None:
        if self._root is None:
            raise RuntimeError("root state is not set")
        e, n = self._engine, int(num_simulations)
        if self._sims_in_tree + n > e.max_sims:
            raise RuntimeError(f"search arena holds {e.max_sims} simulations per root; {self._sims_in_tree} used, {n} requested")
        if not self._expanded:
            kind = int(e.buf["leaf_kind"].item())
            noise = None
            if self.cfg.add_dirichlet_noise:
                self._rng.gamma_into(self._noise, float(self.cfg.dirichlet_alpha), MAX_CHILDREN)
                noise = self._noise
            if kind in (1, 3):                                   # fresh root: evaluate; kept root: fresh noise only
                lp1, lp2, lpm, val = self._evaluate() if kind == 1 else (e.lp1, e.lp2, e.lpm, e.values)
                e.expand(is_root=True, values=val, heads=(lp1, lp2, lpm), noise=noise,
                         epsilon=float(self.cfg.dirichlet_epsilon))
            self._expanded = True
        if int(e.buf["root_terminal"].item()):                   # finished game / no legal move: nothing to search
            return
        for _ in range(n):
            e.select()
            lp1, lp2, lpm, val = self._evaluate()
            e.expand(is_root=False, values=val, heads=(lp1, lp2, lpm))
        self._sims_in_tree += n

    def _children(self) -> Tuple[List[int], List[float], List[float]]:
        e = self._engine
        t = torch.ones((1,), dtype=torch.float32, device=self.device)
        e.finish(t, None)
        k = int(e.child_count.item())
        return (e.child_action[0, :k].tolist(), [float(v) for v in e.child_visits[0, :k].tolist()],
                e.child_prior[0, :k].tolist())

    def get_policy(self, temperature: float = 1.0) -> List[Tuple[int, float]]:
        """mcts_core.cpp:703-760: visits^(1/T) normalised over the root's children (T <= 1e-6: one-hot argmax)."""
        if self._root is None or not self._expanded:
            return []
        acts, visits, _ = self._children()
        if not acts:
            return []
        temp = max(float(temperature), 1e-6)
        if temp <= 1e-6:
            best = max(range(len(visits)), key=lambda i: (visits[i], -i))
            return [(a, 1.0 if i == best else 0.0) for i, a in enumerate(acts)]
        scaled = [v ** (1.0 / temp) for v in visits]
        s = sum(scaled)
        scaled = [1.0 / len(scaled)] * len(scaled) if s <= 0 else [v / s for v in scaled]
        return list(zip(acts, scaled))

    def get_root_children_stats(self) -> List[Dict[str, float]]:
        """mcts_core.cpp:200-217: per child of the root {action_index, prior, visit_count, value_sum}; `value_sum` is
        given from the ROOT mover's side (value_sum / visit_count is the Q the selection rule uses)."""
        if self._root is None or not self._expanded:
            return []
        root_white = int(self._root.current_player.item()) < 0
        out = []
        for rec in self._engine.root_edge_records(0):
            w = rec["value_sum"] if rec["child_white"] == root_white else -rec["value_sum"]
            out.append({"action_index": rec["action_index"], "prior": rec["prior"],
                        "visit_count": float(rec["visit_count"]), "value_sum": w})
        return out

    def advance_root(self, action_index: int) -> None:
        """mcts_core.cpp:815-829: keep the subtree of the played child (tree reuse); unknown child: reset."""
        if self._root is None or not self._expanded:
            return
        from . import v0_core
        acts, _, _ = self._children()
        if int(action_index) not in acts:
            self.reset()
            return
        mask, meta = v0_core.encode_actions_fast(*self._root.tensors()[:10], 36, 144, 36, 4)
        code = meta[0, int(action_index)].view(1, 4).contiguous()
        nxt = GpuStateBatch(*v0_core.batch_apply_moves(*self._root.tensors(), code,
                                                       torch.zeros((1,), dtype=torch.int64, device=self.device)))
        e = self._engine
        e.set_roots(nxt)
        # room for a full arena of new simulations is kept free (a subtree too large for that is dropped: fresh root)
        e.advance(torch.tensor([int(action_index)], dtype=torch.int32, device=self.device), None, e.max_sims)
        self._root, self._root_like = nxt, None
        self._expanded = False
        self._sims_in_tree = 0
        self._rng.ply.add_(1)

    def reset_eval_stats(self) -> None:
        self._eval_calls = self._eval_leaves = 0

    def get_eval_stats(self) -> Dict[str, object]:
        return {"eval_calls": self._eval_calls, "eval_leaves": self._eval_leaves, "full512_calls": 0, "hist": []}

    def get_tree_stats(self) -> Dict[str, int]:
        """Not in the reference (its tree is unbounded): how often advance_root could not keep the whole subtree of the
        played move -- `reuse_pruned`: cut to its oldest part that fits the node arena (statistics of positions past the
        cut restart from zero when they are visited again), `reuse_dropped`: forgotten whole -- and the state of the edge
        pool (`refused_expansions` must stay 0)."""
        if self._engine is None:
            return {"reuse_dropped": 0, "reuse_pruned": 0, "refused_expansions": 0, "node_cap": 0}
        d, p = (int(v) for v in self._engine.reuse_dropped.tolist())
        return {"reuse_dropped": d, "reuse_pruned": p, "node_cap": int(self._engine.node_cap),
                "refused_expansions": int(self._engine.pool_status()["refused_expansions"])}

    @property
    def root_value(self) -> float:
        if self._root is None or self._engine is None:
            return 0.0
        n = int(self._engine.buf["root_visits"].item())
        return float(self._engine.buf["root_w"].item()) / n if n > 0 else 0.0

    @property
    def root_visit_count(self) -> float:
        return 0.0 if self._root is None or self._engine is None else float(self._engine.buf["root_visits"].item())

    @property
    def root_state(self):
        return self._root_like if self._root_like is not None else self._root
