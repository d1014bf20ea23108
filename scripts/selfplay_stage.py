#!/usr/bin/env python3
"""`--stage selfplay` of the reference pipeline on MI355X (scripts/big_train_v1.sh:667-704 ->
scripts/train_entry.py --pipeline v1 --stage selfplay): same flags, same outputs (chunk files + the
`v1_sharded_manifest` at --self_play_output, optional --self_play_stats_json), one worker process per device.

    python scripts/selfplay_stage.py --stage selfplay --devices cuda:0,cuda:1 --self_play_games 32768 \
        --mcts_simulations 800 --self_play_concurrent_games 16384 --self_play_output out/selfplay_iter_001.pt

Flags of the other stages (--train_devices, --batch_size, ...) are accepted and ignored so the reference's command
line can be passed through unchanged.  `--search_backend portable` selects the full-tree engine.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--pipeline", default="v1")
    ap.add_argument("--stage", default="selfplay", choices=["selfplay"])
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--devices", default=None, help="comma-separated self-play devices")
    ap.add_argument("--self_play_games", type=int, default=4)
    ap.add_argument("--mcts_simulations", type=int, default=32)
    ap.add_argument("--temperature_init", type=float, default=1.0)
    ap.add_argument("--temperature_final", type=float, default=0.1)
    ap.add_argument("--temperature_threshold", type=int, default=10)
    ap.add_argument("--exploration_weight", type=float, default=1.0)
    ap.add_argument("--dirichlet_alpha", type=float, default=0.3)
    ap.add_argument("--dirichlet_epsilon", type=float, default=0.25)
    ap.add_argument("--soft_value_k", type=float, default=2.0)
    ap.add_argument("--soft_label_alpha", type=float, default=0.0)
    ap.add_argument("--max_game_plies", type=int, default=512)
    ap.add_argument("--self_play_concurrent_games", type=int, default=8)
    ap.add_argument("--self_play_opening_random_moves", type=int, default=0)
    ap.add_argument("--sparse_ply", type=int, default=1)
    ap.add_argument("--sparse_top_k", type=int, default=8)
    ap.add_argument("--self_play_backend", default="process")
    ap.add_argument("--search_backend", default="cuda_root", choices=["cuda_root", "portable", "tree"])
    ap.add_argument("--policy_target_temperature", type=float, default=None)
    ap.add_argument("--policy_target_prior_pseudocount", type=float, default=0.0)
    ap.add_argument("--self_play_target_samples_per_shard", type=int, default=0)
    ap.add_argument("--self_play_chunk_target_bytes", type=int, default=0)
    ap.add_argument("--self_play_shard_dir", default=None)
    ap.add_argument("--model_init_seed", type=int, default=int(os.environ.get("V1_MODEL_INIT_SEED", "20260314")))
    ap.add_argument("--model", default="b10c128", choices=["b6c64", "b10c128"],
                    help="architecture when no checkpoint is loaded (the reference always builds 10x128)")
    ap.add_argument("--checkpoint_dir", default="./checkpoints_v1")
    ap.add_argument("--load_checkpoint", default=None)
    ap.add_argument("--self_play_output", default=None)
    ap.add_argument("--self_play_iteration_seed", type=int, default=1)
    ap.add_argument("--self_play_stats_json", default=None)
    args, ignored = ap.parse_known_args(argv)
    args.ignored = ignored
    return args


def main(argv=None) -> int:
    args = parse(argv)
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.self_play_stage import run_self_play_stage
    from liuzhou_amd.self_play_worker import _infer_model
    if int(args.self_play_iteration_seed) <= 0:
        raise ValueError(f"self_play_iteration_seed must be positive when provided, got {args.self_play_iteration_seed}")
    devices = [d.strip() for d in str(args.devices or args.device).split(",") if d.strip()]
    if args.load_checkpoint:
        if not os.path.exists(args.load_checkpoint):
            raise FileNotFoundError(f"Checkpoint not found: {args.load_checkpoint}")
        ckpt = torch.load(args.load_checkpoint, map_location="cpu", weights_only=False)
        state = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
        model = _infer_model(state)
        model.load_state_dict(state, strict=True)
    else:
        model = ChessNet(**MODEL_CONFIGS[args.model])
        if int(args.model_init_seed) > 0:
            stable_resnet_init(model, int(args.model_init_seed))
    model.eval()
    output = str(args.self_play_output or os.path.join(args.checkpoint_dir, "selfplay_batch_v1.pt"))
    meta = {"stage": "selfplay", "source_checkpoint": args.load_checkpoint, "self_play_devices": devices,
            "self_play_backend": args.self_play_backend, "search_backend": args.search_backend,
            "self_play_shard_dir": args.self_play_shard_dir, "mcts_simulations": int(args.mcts_simulations),
            "self_play_games": int(args.self_play_games),
            "self_play_concurrent_games": int(args.self_play_concurrent_games),
            "self_play_opening_random_moves": int(args.self_play_opening_random_moves),
            "self_play_iteration_seed": int(args.self_play_iteration_seed),
            "policy_target_temperature": args.policy_target_temperature,
            "policy_target_prior_pseudocount": float(args.policy_target_prior_pseudocount)}
    stats, manifest = run_self_play_stage(
        model_state=model.state_dict(), num_games=int(args.self_play_games), devices=devices, output_path=output,
        iteration_seed=int(args.self_play_iteration_seed), mcts_simulations=int(args.mcts_simulations),
        temperature_init=args.temperature_init, temperature_final=args.temperature_final,
        temperature_threshold=args.temperature_threshold, exploration_weight=args.exploration_weight,
        dirichlet_alpha=args.dirichlet_alpha, dirichlet_epsilon=args.dirichlet_epsilon, soft_value_k=args.soft_value_k,
        soft_label_alpha=args.soft_label_alpha, opening_random_moves=args.self_play_opening_random_moves,
        max_game_plies=args.max_game_plies, concurrent_games_per_device=args.self_play_concurrent_games,
        shard_dir=args.self_play_shard_dir, target_samples_per_shard=args.self_play_target_samples_per_shard,
        chunk_target_bytes=args.self_play_chunk_target_bytes, metadata_base=meta, sparse_ply=args.sparse_ply,
        sparse_top_k=args.sparse_top_k, search_backend=args.search_backend,
        policy_target_temperature=args.policy_target_temperature,
        policy_target_prior_pseudocount=args.policy_target_prior_pseudocount)
    print(f"[selfplay] games={stats.num_games} positions={stats.num_positions} "
          f"positions/s={stats.positions_per_sec:.1f} W/L/D={stats.black_wins}/{stats.white_wins}/{stats.draws} "
          f"shards={manifest['num_shards']} -> {output}", flush=True)
    if args.self_play_stats_json:
        os.makedirs(os.path.dirname(args.self_play_stats_json) or ".", exist_ok=True)
        with open(args.self_play_stats_json, "w") as f:
            json.dump({"stats": stats.to_dict(), "num_shards": manifest["num_shards"],
                       "num_samples": manifest["num_samples"], "output": output,
                       "value_target_summary": manifest["metadata"].get("value_target_summary", {})}, f, indent=2)
    return 0


if __name__ == "__main__":
    sys.exit(main())
