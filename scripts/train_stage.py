#!/usr/bin/env python3
"""`--stage train` of the reference pipeline on MI355X (scripts/big_train_v1.sh:780-848 ->
scripts/train_entry.py --pipeline v1 --stage train): same flags, reads the self-play manifest (+ replay window),
trains with the fused loss kernel, writes `<checkpoint_dir>/<checkpoint_name>` ({"model_state_dict": ...}) and the
metrics json.  Under torchrun (`--train_strategy ddp`) every rank trains on its `rank::world` share of the shards.

    python scripts/train_stage.py --stage train --self_play_input runs/selfplay_iter_001.pt --streaming_load 1 \
        --batch_size 4096 --epochs 2 --lr 1e-3 --checkpoint_dir ck --checkpoint_name model_iter_001.pt \
        --metrics_output runs/train_iter_001.json
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


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--pipeline", default="v1")
    ap.add_argument("--stage", default="train", choices=["train"])
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--train_strategy", default="none")
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--weight_decay", type=float, default=1e-4)
    ap.add_argument("--soft_label_alpha", type=float, default=0.0)
    ap.add_argument("--anti_draw_penalty", type=float, default=0.0)
    ap.add_argument("--policy_draw_weight", type=float, default=1.0)
    ap.add_argument("--warmup_steps", type=int, default=0)
    ap.add_argument("--checkpoint_dir", default="./checkpoints_v1")
    ap.add_argument("--checkpoint_name", default=None)
    ap.add_argument("--load_checkpoint", default=None)
    ap.add_argument("--self_play_input", required=True)
    ap.add_argument("--self_play_replay_inputs", default=None)
    ap.add_argument("--replay_budget_per_file", type=int, default=0)
    ap.add_argument("--optimizer_state_path", default=None)
    ap.add_argument("--streaming_load", type=int, default=0)
    ap.add_argument("--streaming_workers", type=int, default=8)
    ap.add_argument("--model_init_seed", type=int, default=int(os.environ.get("V1_MODEL_INIT_SEED", "20260314")))
    ap.add_argument("--model", default="b10c128", choices=["b6c64", "b10c128"])
    ap.add_argument("--metrics_output", default=None)
    args, ignored = ap.parse_known_args(argv)
    args.ignored = ignored
    return args


def main(argv=None) -> int:
    args = parse(argv)
    from liuzhou_amd import self_play_stage as S
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.self_play_worker import _infer_model
    from liuzhou_amd.streaming import build_streaming_dataloader
    from liuzhou_amd.train_bridge import train_network_from_tensors, train_network_streaming
    ddp = str(args.train_strategy).strip().lower() == "ddp" and "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank, world = 0, 1
    device = args.device
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        rank, world, device = dist.get_rank(), dist.get_world_size(), f"cuda:{local}"
    strategy = "ddp" if ddp else "none"
    t0 = time.perf_counter()
    if args.load_checkpoint:
        ck = torch.load(args.load_checkpoint, map_location="cpu", weights_only=False)
        state = ck["model_state_dict"] if isinstance(ck, dict) and "model_state_dict" in ck else ck
        model = _infer_model(state)
        model.load_state_dict(state, strict=True)
    else:
        model = ChessNet(**MODEL_CONFIGS[args.model])
        if int(args.model_init_seed) > 0:
            stable_resnet_init(model, int(args.model_init_seed))
    load_sec = time.perf_counter() - t0
    replay = [p.strip() for p in str(args.self_play_replay_inputs or "").split(",") if p.strip()]
    common = dict(batch_size=args.batch_size, epochs=args.epochs, lr=args.lr, weight_decay=args.weight_decay,
                  soft_label_alpha=args.soft_label_alpha, anti_draw_penalty=args.anti_draw_penalty,
                  policy_draw_weight=args.policy_draw_weight, device=device, warmup_steps=args.warmup_steps,
                  parallel_strategy=strategy, optimizer_state_path=args.optimizer_state_path)
    t0 = time.perf_counter()
    if int(args.streaming_load):
        budget = int(args.replay_budget_per_file)
        if replay and budget <= 0:                     # v1/train.py:2367: the replay window weighs as much as the new data
            _, primary_est = S.resolve_shard_specs(args.self_play_input, [], 0, ddp_rank=rank, ddp_world=world)
            budget = max(1, primary_est // len(replay))
        specs, total = S.resolve_shard_specs(args.self_play_input, replay, budget, ddp_rank=rank, ddp_world=world)
        loader = build_streaming_dataloader(specs, batch_size=args.batch_size, num_workers=args.streaming_workers,
                                            epoch_seed=rank)
        model, metrics = train_network_streaming(model, loader, total_samples=total, streaming_workers=args.streaming_workers,
                                                 **common)
    else:
        batches = [S.load_self_play_payload(p, ddp_rank=rank if ddp else None, ddp_world_size=world if ddp else None)[0]
                   for p in [args.self_play_input] + [r for r in replay if os.path.exists(r)]]
        samples = S.concat_batches(batches)
        model, metrics = train_network_from_tensors(model, samples, ddp_pre_sharded=True, **common)
    train_sec = time.perf_counter() - t0
    if ddp:
        dist.barrier()
    if rank == 0:
        os.makedirs(args.checkpoint_dir, exist_ok=True)
        ckpt = os.path.join(args.checkpoint_dir, str(args.checkpoint_name or "model_iter_001.pt"))
        torch.save({"iteration": 1, "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "board_size": 6, "num_input_channels": 11, "train_strategy": strategy, "stage": "train",
                    "self_play_input": str(args.self_play_input)}, ckpt)
        last = (metrics.get("epoch_stats") or [{}])[-1]
        entry = {"stage": "train", "streaming": bool(int(args.streaming_load)), "self_play_input": str(args.self_play_input),
                 "train_strategy": strategy, "checkpoint_load_sec": load_sec, "train_time_sec": train_sec,
                 "train_avg_loss": last.get("avg_loss"), "train_avg_policy_loss": last.get("avg_policy_loss"),
                 "train_avg_value_loss": last.get("avg_value_loss"), "train_soft_label_alpha": float(args.soft_label_alpha),
                 "checkpoint": ckpt, "train_bridge": metrics}
        print(f"[train] samples={last.get('samples')} loss={last.get('avg_loss')} policy={last.get('avg_policy_loss')} "
              f"value={last.get('avg_value_loss')} time={train_sec:.1f}s -> {ckpt}", flush=True)
        if args.metrics_output:
            os.makedirs(os.path.dirname(args.metrics_output) or ".", exist_ok=True)
            with open(args.metrics_output, "w") as f:
                json.dump([entry], f, indent=2, default=str)
    if ddp:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
