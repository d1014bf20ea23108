#!/usr/bin/env python3
"""Staged self-play / training loop of BASELINE.json configs[4] (C5): self-play on ranks 0..N-2, trainer on rank N-1,
trajectories gathered to the trainer as compact records over RCCL, checkpoint broadcast back every iteration
(the loop `scripts/big_train_v1.sh:647-821` runs as shell stages over files; `v1/train.py:932-1020` shards the games).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/staged_loop.py \
        --games-per-gpu 16384 --sims 800 --model b10c128 --search tree --iterations 3 --overlap 1

Schedules (one iteration = one generation of games on every player):
  --overlap 0   the reference's order: play(i) -> gather(i) -> train(i) -> broadcast(i).  Players idle while the trainer
                trains, the trainer idles while they play.
  --overlap 1   lag-1 pipeline: while the players play generation i (on the weights trained on generation i-2), the
                trainer trains on generation i-1 (gathered at the end of the previous iteration); at the end of the
                iteration generation i is gathered and the new weights are broadcast.  An iteration costs
                max(play, train) + gather + broadcast instead of their sum; the data a checkpoint has seen is one
                generation older than in the sequential order (the reference's replay window mixes several past
                generations anyway, `v1/train.py:1486-1735`).  After the last iteration the trainer trains on the
                last generation ("tail"), so both schedules end with the same number of training passes.

With one process everything runs on the same GPU (self-play, then training; --overlap has nothing to overlap).
Tests rehearse the multi-rank branches with several ranks on ONE GPU: `LZ_DIST_BACKEND=gloo LZ_SHARE_GPU=1` (RCCL
refuses two ranks per device; the gather then stages its records through host memory, same protocol and kernels).
One JSON line per run (printed by the trainer rank).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist


def _digest_pack(wfrag: torch.Tensor, fparams: torch.Tensor) -> int:
    """Exact integer digest of a packed weight set (fp16 fragments + fp32 parameters read as integers): equal on two
    ranks iff the kernels of both would read the same bits."""
    a = wfrag.contiguous().view(torch.int16).to(torch.int64)
    b = fparams.contiguous().view(torch.int32).to(torch.int64)
    ia = torch.arange(1, a.numel() + 1, dtype=torch.int64, device=a.device)
    ib = torch.arange(1, b.numel() + 1, dtype=torch.int64, device=b.device)
    return int(((a * (ia % 8191 + 1)).sum() + (b % 1000003 * (ib % 8191 + 1)).sum()).item())


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--games-per-gpu", type=int, default=256)
    ap.add_argument("--sims", type=int, default=32)
    ap.add_argument("--model", default="b6c64", choices=("b6c64", "b10c128"))
    ap.add_argument("--search", default="tree", choices=("tree", "root"))
    ap.add_argument("--max-game-plies", type=int, default=512)
    ap.add_argument("--batch-size", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--soft-label-alpha", type=float, default=0.3)
    ap.add_argument("--overlap", type=int, default=0, choices=(0, 1),
                    help="1: lag-1 pipeline, the trainer trains on generation i-1 while generation i is played")
    ap.add_argument("--net-check", type=int, default=0,
                    help="single process: after every checkpoint hand-off evaluate this many of the iteration's positions "
                         "with the refreshed fp16 kernel, the fp32 module and torch.autocast(float16) and log the differences")
    ap.add_argument("--check-digests", type=int, default=1,
                    help="after every hand-off compare an exact digest of each player's packed device weights with the "
                         "trainer's (outside the timed part of the iteration)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LZ_DIST_BACKEND", "nccl")
    share = os.environ.get("LZ_SHARE_GPU", "0") == "1"
    dev_index = local % max(1, torch.cuda.device_count()) if share else local
    dev = torch.device(f"cuda:{dev_index}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    trainer = world - 1
    plays = world == 1 or rank != trainer
    trains = rank == trainer
    overlap = bool(args.overlap) and world > 1

    from liuzhou_amd.distributed import broadcast_checkpoint, gather_trajectories, worker_seed
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.net_pack import pack_model
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    from liuzhou_amd.train_bridge import train_network_from_tensors
    from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
    from liuzhou_amd.tree_engine import self_play_tree_gpu

    def reduce_max(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        return float(t.item())

    def all_digests(mine: int):
        if world == 1:
            return [mine]
        t = torch.tensor([mine], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        return [int(e.item()) for e in every]

    def train(model, data):
        model, metrics = train_network_from_tensors(model, data, batch_size=args.batch_size, epochs=args.epochs,
                                                    lr=args.lr, soft_label_alpha=args.soft_label_alpha, device=str(dev))
        model.eval()
        e = metrics["epoch_stats"][-1] if metrics and metrics.get("epoch_stats") else {}
        # rows handed to the trainer; `samples_stepped` (epoch_stats.samples) can be lower: the AMP scaler skips the
        # batches whose scaled gradients overflow, exactly as the reference's loop does (train_bridge.py:388-420)
        return model, int(metrics.get("num_samples", 0)), {"avg_loss": e.get("avg_loss"), "samples_stepped": int(e.get("samples", 0))}

    model = ChessNet(**MODEL_CONFIGS[args.model])
    stable_resnet_init(model, 20260314)                               # MODEL_INIT_SEED (big_train_v1.sh:24)
    model.to(dev).eval()
    log = []
    fused = FusedNet(model, dev) if plays else None
    pending = None                                                     # overlap: the generation gathered last iteration
    digests_seen = []
    for it in range(1, args.iterations + 1):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        t_play = t_train = 0.0
        trained, loss = 0, None
        if plays:
            torch.manual_seed(worker_seed(it, rank))                   # v1/train.py:998
            play = self_play_tree_gpu if args.search == "tree" else self_play_v1_gpu
            batch, _stats = play(fused, num_games=args.games_per_gpu, mcts_simulations=args.sims,
                                 temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
                                 exploration_weight=1.0, device=str(dev), max_game_plies=args.max_game_plies,
                                 concurrent_games=args.games_per_gpu)
            torch.cuda.synchronize(dev)
            t_play = time.perf_counter() - t0
        else:
            z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
            batch = TensorSelfPlayBatch(z(0, 11, 6, 6), z(0, 220, dt=torch.bool), z(0, 220), z(0), z(0))
            if overlap and pending is not None:                        # generation it-1, while the players play `it`
                model, trained, loss = train(model, pending)
                pending = None
                torch.cuda.synchronize(dev)
                t_train = time.perf_counter() - t0
        t1 = time.perf_counter()
        gathered = gather_trajectories(batch, dst=trainer) if world > 1 else batch
        torch.cuda.synchronize(dev)
        t_gather = time.perf_counter() - t1
        if trains:
            if overlap:
                pending = gathered
            else:
                t2 = time.perf_counter()
                model, trained, loss = train(model, gathered)
                torch.cuda.synchronize(dev)
                t_train = time.perf_counter() - t2
        t3 = time.perf_counter()
        broadcast_checkpoint(model, src=trainer)
        if fused is not None:
            fused.refresh(model)                                       # new checkpoint into the same device buffers
        torch.cuda.synchronize(dev)
        t_handoff = time.perf_counter() - t3
        net_check = None
        if args.net_check > 0 and world == 1 and fused is not None and batch.num_samples > 0:
            # the TRAINED weights in the fp16-activation kernel: against the fp32 module and against the reference's own
            # inference mode (torch.autocast(float16), v1/python/mcts_gpu.py:640-646) on positions of this iteration
            x = batch.state_tensors[: int(args.net_check)].contiguous()
            with torch.inference_mode():
                r32 = model(x)
                with torch.autocast("cuda", dtype=torch.float16):
                    r16 = tuple(t.float() for t in model(x))
            f = fused(x)
            fv = fused.last_value
            from liuzhou_amd.net import bucket_logits_to_scalar
            v32, v16 = bucket_logits_to_scalar(r32[3]), bucket_logits_to_scalar(r16[3])
            finite = all(bool(torch.isfinite(t).all()) for t in f) and bool(torch.isfinite(fv).all())
            d_f = max(float((f[k].exp() - r32[k].exp()).abs().max()) for k in range(3))
            d_a = max(float((r16[k].exp() - r32[k].exp()).abs().max()) for k in range(3))
            agree = min(float((f[k].argmax(1) == r32[k].argmax(1)).float().mean()) for k in range(3))
            net_check = {"positions": int(x.shape[0]), "finite": finite, "max_dprob_fused_vs_fp32": d_f,
                         "max_dprob_autocast_vs_fp32": d_a, "argmax_agreement_min_head": agree,
                         "max_dvalue_fused_vs_fp32": float((fv - v32).abs().max()),
                         "max_dvalue_autocast_vs_fp32": float((v16 - v32).abs().max()),
                         "max_abs_logprob_fp32": max(float(r32[k].abs().max()) for k in range(3)),
                         "policy_entropy_mean_fp32": float(-(r32[0].exp() * r32[0]).sum(1).mean())}
        if world > 1:
            dist.barrier()
        t_total = time.perf_counter() - t0
        n = reduce_sum(float(batch.num_samples))
        play_max, gather_max = reduce_max(t_play), reduce_max(t_gather)
        same = None
        if args.check_digests:
            if fused is not None:
                mine = _digest_pack(fused.pack.wfrag, fused.pack.fparams)
            else:
                p = pack_model(model)
                mine = _digest_pack(p.wfrag, p.fparams)
            every = all_digests(mine)
            same = len(set(every)) == 1
            digests_seen.append(every[trainer])
        if trains:
            log.append({"iteration": it, "positions": int(n), "self_play_sec": round(play_max, 3),
                        "gather_sec": round(gather_max, 3), "train_sec": round(t_train, 3),
                        "handoff_sec": round(t_handoff, 3), "iteration_sec": round(t_total, 3),
                        "positions_per_sec": round(n / max(t_total, 1e-9), 1),
                        "train_samples": trained, "avg_loss": (loss or {}).get("avg_loss"),
                        "samples_stepped": (loss or {}).get("samples_stepped"), "weights_equal_on_all_ranks": same,
                        "weights_digest": digests_seen[-1] if digests_seen else None, "net_check": net_check})
    tail = None
    if trains and overlap and pending is not None:                     # the last generation, nothing left to overlap with
        t0 = time.perf_counter()
        model, trained, loss = train(model, pending)
        torch.cuda.synchronize(dev)
        tail = {"train_samples": trained, "avg_loss": (loss or {}).get("avg_loss"),
                "samples_stepped": (loss or {}).get("samples_stepped"), "train_sec": round(time.perf_counter() - t0, 3)}
    if trains:
        steady = log[1:] if len(log) > 1 else log
        out = {"workload": f"C5 staged loop: {max(1, world - 1)} self-play GPU(s) x {args.games_per_gpu} games, "
                           f"{args.sims} sims/move, {args.model}, search={args.search}; trainer on rank {trainer}; "
                           f"{'lag-1 overlap of training and self-play' if overlap else 'sequential'}",
               "world_size": world, "backend": (backend if world > 1 else None), "overlap": int(overlap),
               "player_ranks": [r for r in range(world) if world == 1 or r != trainer], "trainer_rank": trainer,
               "iterations": log, "tail": tail,
               "steady_state_positions_per_sec": round(sum(x["positions"] for x in steady) /
                                                       max(1e-9, sum(x["iteration_sec"] for x in steady)), 1)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
