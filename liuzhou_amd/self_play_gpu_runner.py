"""Wave loop of v1 self-play on MI355X (mirror of v1/python/self_play_gpu_runner.py:21-307).

`self_play_v1_gpu(**kwargs)` keeps the reference's 22 parameters / defaults and returns
`(TensorSelfPlayBatch, SelfPlayV1Stats)`.  All active games of a wave are searched and stepped together;
per ply: search -> append trajectory rows -> `self_play_step_inplace` -> finalize finished games.
"""
from __future__ import annotations

import os
import time
from typing import Dict, Tuple

import torch

from . import v0_core
from .mcts_gpu import GpuStateBatch, TOTAL_ACTION_DIM, V1RootMCTS, V1RootMCTSConfig
from .self_play_types import SelfPlayV1Stats
from .trajectory_buffer import TensorSelfPlayBatch, TensorTrajectoryBuffer

_DELTA_MIN, _DELTA_MAX = -18, 18
_TRACKED = ("root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms")


def streaming_supported(model, *, opening_random_moves: int = 0, child_eval_mode: str = "value_only", sparse_ply: int = 1,
                        inference_engine=None, collect_step_timing: bool = False) -> bool:
    """True when `self_play_v1_gpu` would take the fused search + device tail for these options, i.e. when it can feed a
    finished-row log (`row_log`).  (`opening_random_moves` is covered by the fused search since round 5 -- the reference's
    training script starts with 6, scripts/big_train_v1.sh:42; `sparse_ply` > 1 and `child_eval_mode="full"` since round 6:
    extra fixed-shape stages of the fused search, so no reference option leaves the sync-free path any more.)"""
    return (hasattr(model, "desc") and inference_engine is None and
            str(child_eval_mode).strip().lower() in ("value_only", "full") and int(sparse_ply) >= 1 and
            not collect_step_timing and os.environ.get("LZ_WAVE_TAIL", "1") != "0")


def self_play_v1_gpu(model, num_games: int, mcts_simulations: int, temperature_init: float,
                     temperature_final: float, temperature_threshold: int, exploration_weight: float, device: str,
                     add_dirichlet_noise: bool = True, dirichlet_alpha: float = 0.3, dirichlet_epsilon: float = 0.25,
                     soft_value_k: float = 2.0, opening_random_moves: int = 0, max_game_plies: int = 512,
                     sample_moves: bool = True, concurrent_games: int = 8, child_eval_mode: str = "value_only",
                     sparse_ply: int = 1, sparse_top_k: int = 8, inference_engine=None,
                     collect_step_timing: bool = False, verbose: bool = False,
                     autocast_dtype: str = "float16", fused_search: bool = True, continuous_waves: bool = True,
                     row_log=None) -> Tuple[TensorSelfPlayBatch, SelfPlayV1Stats]:
    """`row_log` (finished_log.FinishedRowLog, the streaming worker): the rows of every game leave through the log when
    the game ends and the returned batch is empty; needs the fused search with the device tail and continuous waves
    (`streaming_supported`)."""
    if num_games <= 0:
        raise ValueError("num_games must be positive.")
    dev = torch.device(device)
    max_plies = max(1, int(max_game_plies))
    opening_n = max(0, int(opening_random_moves))
    wave = max(1, min(int(concurrent_games), int(num_games)))
    cfg = V1RootMCTSConfig(
        num_simulations=max(1, int(mcts_simulations)), exploration_weight=float(exploration_weight),
        temperature=float(temperature_init), add_dirichlet_noise=bool(add_dirichlet_noise),
        dirichlet_alpha=float(dirichlet_alpha), dirichlet_epsilon=float(dirichlet_epsilon),
        sample_moves=bool(sample_moves), child_eval_mode=str(child_eval_mode), soft_value_k=float(soft_value_k),
        sparse_ply=max(1, int(sparse_ply)), sparse_top_k=max(1, int(sparse_top_k)), autocast_dtype=autocast_dtype)
    mcts = V1RootMCTS(model=model, config=cfg, device=dev, inference_engine=inference_engine,
                      collect_timing=bool(collect_step_timing))
    buffer = TensorTrajectoryBuffer(dev, TOTAL_ACTION_DIM, max_steps_hint=max_plies, concurrent_games_hint=wave)
    # Fused search (root_search_fused.py): whole waves through one captured, sync-free launch sequence.  It needs the
    # fused network, a fixed batch (finished games of the wave are searched too and their rows dropped) and none of
    # the options only the operator chain implements.
    fused = None
    if (fused_search and hasattr(model, "desc") and inference_engine is None and
            str(child_eval_mode).strip().lower() in ("value_only", "full") and not collect_step_timing and
            (int(num_games) % wave == 0 or continuous_waves)):
        from .root_search_fused import FusedRootSearch
        # The reference draws noise and moves from the process's torch generator, which its worker seeds per shard
        # (v1/python/self_play_worker.py:300-303 `torch.manual_seed(seed)`); the per-game counter RNG here takes its key
        # from that generator, so workers / chunks seeded differently play different games and a seed reproduces its games
        rng_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        fused = FusedRootSearch(model, wave, cfg.num_simulations, dev, seed=rng_seed, exploration_weight=cfg.exploration_weight,
                                add_dirichlet_noise=cfg.add_dirichlet_noise, dirichlet_alpha=cfg.dirichlet_alpha,
                                dirichlet_epsilon=cfg.dirichlet_epsilon, sample_moves=cfg.sample_moves,
                                soft_value_k=cfg.soft_value_k, sparse_ply=cfg.sparse_ply, sparse_top_k=cfg.sparse_top_k,
                                child_eval_mode=cfg.child_eval_mode)

    outcome = torch.zeros((3,), dtype=torch.int64, device=dev)
    delta_hist = torch.zeros((_DELTA_MAX - _DELTA_MIN + 1,), dtype=torch.int64, device=dev)
    tail = None
    if fused is not None and os.environ.get("LZ_WAVE_TAIL", "1") != "0":   # device-side record / move / finalise (wave_tail.py)
        from .wave_tail import WaveTail
        tail = WaveTail(buffer, wave, max_plies, dev, soft_value_k=float(soft_value_k), row_log=row_log)
        outcome, delta_hist = tail.outcome, tail.delta_hist
    if row_log is not None and not (tail is not None and bool(continuous_waves)):
        raise RuntimeError("self_play_v1_gpu: a finished-row log needs the fused search, the device tail and continuous "
                           "waves (fused network, no external inference engine, no step timing)")
    lengths = torch.zeros((int(num_games),), dtype=torch.int64, device=dev)
    timing_ms: Dict[str, float] = {k: 0.0 for k in _TRACKED}
    timing_calls: Dict[str, int] = {k: 0 for k in _TRACKED}
    events = []

    class _timed:
        def __init__(self, name):
            self.n = name

        def __enter__(self):
            if collect_step_timing:
                self.s = torch.cuda.Event(enable_timing=True); self.e = torch.cuda.Event(enable_timing=True)
                self.s.record()

        def __exit__(self, *a):
            if collect_step_timing:
                self.e.record(); events.append((self.n, self.s, self.e))

    started = time.perf_counter()
    # continuous_waves (fused path): ONE wave whose finished slots start the remaining games at once, instead of the
    # reference's sequential waves that idle until their longest game has ended (self_play_gpu_runner.py:84-90)
    continuous = tail is not None and bool(continuous_waves)
    for base in range(0, wave if continuous else int(num_games), wave):
        g = min(wave, int(num_games) - base)
        states = GpuStateBatch.initial(dev, g)
        step_index = None if row_log is not None else torch.full((g, max_plies), -1, dtype=torch.int64, device=dev)
        step_counts = torch.zeros((g,), dtype=torch.int64, device=dev)
        plies = torch.zeros((g,), dtype=torch.int64, device=dev)
        done = torch.zeros((g,), dtype=torch.bool, device=dev)
        ones = torch.ones((g,), dtype=torch.int64, device=dev)
        if tail is not None:
            # reset / game ids / plies: the runner's own numbering keys the per-game RNG (noise, picks, opening moves), so a
            # game's draws do not depend on the slot or the moment it is played
            tail.run(lambda st, temps, dn, reseated: fused.search_batch(
                         st, temperatures=temps, add_dirichlet_noise=add_dirichlet_noise,
                         rng_game_ids=tail.slot_game + (0 if continuous else base), rng_plies=plies,
                         force_uniform_random_mask=(plies < opening_n) if opening_n > 0 else None),
                     states, plies, done, step_index, step_counts, lengths if continuous else lengths[base:base + g],
                     temperature_init, temperature_final, temperature_threshold,
                     games_to_start=int(num_games) - wave if continuous else 0)
            continue
        while True:
            active = torch.nonzero(~done).view(-1)
            n_active = int(active.numel())
            if n_active == 0:
                break
            act_states = states if n_active == g else states.select(active)
            act_plies = plies.index_select(0, active)
            temps = torch.where(act_plies < int(temperature_threshold), float(temperature_init),
                                float(temperature_final)).to(torch.float32)
            force = (act_plies < opening_n) if opening_n > 0 else None
            if fused is not None and g == fused.B:              # LZ_WAVE_TAIL=0: fused search, host-side tail
                full_temps = torch.where(plies < int(temperature_threshold), float(temperature_init),
                                         float(temperature_final)).to(torch.float32)
                search = fused.search_batch(states, temperatures=full_temps, add_dirichlet_noise=add_dirichlet_noise)
                if n_active != g:
                    search = type(search)(*(getattr(search, f).index_select(0, active) for f in (
                        "model_input", "legal_mask", "policy_dense", "root_value", "terminal_mask",
                        "chosen_action_indices", "chosen_action_codes", "chosen_valid_mask")))
            else:
                search = mcts.search_batch(act_states, temperatures=temps, add_dirichlet_noise=add_dirichlet_noise,
                                           force_uniform_random_mask=force)
            rows = buffer.append_steps(search.model_input, search.legal_mask, search.policy_dense,
                                       act_states.current_player)
            step_index[active, step_counts.index_select(0, active)] = rows
            step_counts.index_add_(0, active, ones[:n_active])
            with _timed("self_play_step_ms"):
                fin_slots, result, _soft = v0_core.self_play_step_inplace(
                    *states.tensors(), plies, done, active, search.chosen_action_codes, search.terminal_mask,
                    search.chosen_valid_mask, int(max_plies), float(soft_value_k))
            if int(fin_slots.numel()) > 0:
                boards = states.board.index_select(0, fin_slots)
                black = boards.eq(1).sum(dim=(1, 2)); white = boards.eq(-1).sum(dim=(1, 2))
                bucket = (black - white - _DELTA_MIN).clamp(0, _DELTA_MAX - _DELTA_MIN)
                delta_hist.add_(torch.bincount(bucket, minlength=_DELTA_MAX - _DELTA_MIN + 1))
                soft = V1RootMCTS._soft_tanh_from_board_black(boards, float(soft_value_k))
                with _timed("finalize_ms"):
                    f_slots, f_len, f_out = buffer.finalize_games_inplace(
                        step_index_matrix=step_index, step_counts=step_counts, slots=fin_slots,
                        result_from_black=result, soft_value_from_black=soft)
                if int(f_slots.numel()) > 0:
                    lengths.index_copy_(0, f_slots + base, f_len)
                outcome.add_(f_out)
        if verbose:
            o = outcome.tolist()
            print(f"[v1.self_play] games={min(base + g, num_games)}/{num_games} W/L/D={o[0]}/{o[1]}/{o[2]}")

    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if tail is not None:
        tail.check_overflow()
    if row_log is not None:
        row_log.close()                                      # the last segment leaves; the rows are the consumer's now
    elapsed = max(1e-9, time.perf_counter() - started)
    for name, s, e in events:
        timing_ms[name] += float(s.elapsed_time(e)); timing_calls[name] += 1
    batch = buffer.build()
    mt = mcts.get_timing(reset=False)
    for k, v in mt["timing_ms"].items():
        timing_ms[k] = timing_ms.get(k, 0.0) + float(v)
    for k, v in mt["timing_calls"].items():
        timing_calls[k] = timing_calls.get(k, 0) + int(v)
    total = sum(timing_ms[k] for k in _TRACKED)
    o = outcome.tolist()
    hist = delta_hist.tolist()
    positions = int(lengths.sum().item()) if row_log is not None else batch.num_samples
    stats = SelfPlayV1Stats(
        num_games=num_games, num_positions=positions, black_wins=int(o[0]), white_wins=int(o[1]),
        draws=int(o[2]), avg_game_length=float(lengths.to(torch.float32).mean().item()), elapsed_sec=elapsed,
        positions_per_sec=float(positions / elapsed), games_per_sec=float(num_games / elapsed),
        step_timing_ms={k: float(timing_ms[k]) for k in _TRACKED},
        step_timing_ratio={k: (float(timing_ms[k]) / total if total > 0 else 0.0) for k in _TRACKED},
        step_timing_calls={k: int(timing_calls[k]) for k in _TRACKED},
        mcts_counters={**{k: int(v) for k, v in mt["counters"].items()},
                       **({"leaf_eval_count": int(fused.leaf_evals), "fused_root_search": 1} if fused is not None else {}),
                       **({"loop_ms": int(tail.loop_ms), "host_wait_ms": int(tail.host_wait_ms),
                           "plies_launched": int(tail.plies_launched)} if tail is not None else {})},
        piece_delta_buckets={str(d): int(hist[d - _DELTA_MIN]) for d in range(_DELTA_MIN, _DELTA_MAX + 1)},
        device=str(dev))
    return batch, stats
