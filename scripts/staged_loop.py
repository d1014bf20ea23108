#!/usr/bin/env python3
"""Staged self-play / training loop of BASELINE.json configs[4] (C5): self-play on ranks 0..N-2, trainer on rank N-1,
trajectories gathered to the trainer as compact records over RCCL, checkpoint broadcast back every iteration.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/staged_loop.py \
        --games-per-gpu 16384 --sims 800 --model b10c128 --search tree --iterations 3

With one process everything runs on the same GPU (self-play, then training).  One JSON line per run (rank 0).
The pieces are the ones of the hot path and its neighbours: `self_play_tree_gpu` / `self_play_v1_gpu`,
`distributed.gather_trajectories` + `broadcast_checkpoint`, `train_bridge.train_network_from_tensors`.
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
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    trainer = world - 1
    plays = world == 1 or rank != trainer

    from liuzhou_amd.distributed import broadcast_checkpoint, gather_trajectories, worker_seed
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    from liuzhou_amd.train_bridge import train_network_from_tensors
    from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
    from liuzhou_amd.tree_engine import self_play_tree_gpu

    model = ChessNet(**MODEL_CONFIGS[args.model])
    stable_resnet_init(model, 20260314)                               # MODEL_INIT_SEED (big_train_v1.sh:24)
    model.to(dev).eval()
    log = []
    fused = FusedNet(model, dev) if plays else None
    for it in range(1, args.iterations + 1):
        if fused is not None and it > 1:
            fused.refresh(model)                                       # new checkpoint into the same device buffers
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if plays:
            torch.manual_seed(worker_seed(it, rank))                   # v1/train.py:998
            play = self_play_tree_gpu if args.search == "tree" else self_play_v1_gpu
            batch, stats = play(fused, num_games=args.games_per_gpu, mcts_simulations=args.sims,
                                temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
                                exploration_weight=1.0, device=str(dev), max_game_plies=args.max_game_plies,
                                concurrent_games=args.games_per_gpu)
        else:
            z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
            batch = TensorSelfPlayBatch(z(0, 11, 6, 6), z(0, 220, dt=torch.bool), z(0, 220), z(0), z(0))
        torch.cuda.synchronize(dev)
        t_play = time.perf_counter() - t0
        gathered = gather_trajectories(batch, dst=trainer) if world > 1 else batch
        torch.cuda.synchronize(dev)
        t_gather = time.perf_counter() - t0 - t_play
        metrics = None
        if rank == trainer:
            model, metrics = train_network_from_tensors(model, gathered, batch_size=args.batch_size, epochs=args.epochs,
                                                        lr=args.lr, soft_label_alpha=args.soft_label_alpha, device=str(dev))
            model.eval()
        broadcast_checkpoint(model, src=trainer)
        torch.cuda.synchronize(dev)
        t_total = time.perf_counter() - t0
        n = torch.tensor([float(batch.num_samples)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(n)
        if rank == trainer:
            e = metrics["epoch_stats"][-1] if metrics and metrics["epoch_stats"] else {}
            log.append({"iteration": it, "positions": int(n.item()), "self_play_sec": round(t_play, 3),
                        "gather_sec": round(t_gather, 3), "iteration_sec": round(t_total, 3),
                        "positions_per_sec": round(float(n.item()) / max(t_total, 1e-9), 1),
                        "train_samples": int(e.get("samples", 0)), "avg_loss": e.get("avg_loss")})
    if rank == trainer:
        steady = log[1:] if len(log) > 1 else log
        out = {"workload": f"C5 staged loop: {max(1, world - 1)} self-play GPU(s) x {args.games_per_gpu} games, "
                           f"{args.sims} sims/move, {args.model}, search={args.search}; trainer on rank {trainer}",
               "world_size": world, "iterations": log,
               "steady_state_positions_per_sec": round(sum(x["positions"] for x in steady) /
                                                       max(1e-9, sum(x["iteration_sec"] for x in steady)), 1)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
