#!/usr/bin/env python3
"""Checkpoint evaluation on the device-resident arena: the `--backend v1` path of the reference's
`scripts/eval_checkpoint.py` (flags of :831-872; other backends' flags are accepted and ignored).

    python scripts/eval_arena.py --challenger_checkpoint ck/model_iter_003.pt --previous_checkpoint ck/best.pt \
        --eval_games_vs_random 200 --eval_games_vs_previous 400 --mcts_simulations 256 --output_json out/eval.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate checkpoint against random/previous.")
    ap.add_argument("--challenger_checkpoint", required=True)
    ap.add_argument("--previous_checkpoint", default=None)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--backend", default="v1")
    ap.add_argument("--mcts_simulations", type=int, default=256)
    ap.add_argument("--temperature", type=float, default=0.05)
    ap.add_argument("--sample_moves", action="store_true")
    ap.add_argument("--eval_games_vs_random", type=int, default=0)
    ap.add_argument("--eval_games_vs_previous", type=int, default=0)
    ap.add_argument("--v1_opening_random_moves", type=int, default=0)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--match_name", default=None)
    ap.add_argument("--output_json", default=None)
    args, ignored = ap.parse_known_args(argv)
    args.ignored = ignored
    return args


def main(argv=None) -> int:
    args = parse(argv)
    from liuzhou_amd.eval_arena import evaluate_checkpoint
    seed = 0 if args.seed is None else int(args.seed)
    common = dict(device=args.device, mcts_simulations=args.mcts_simulations, temperature=args.temperature,
                  sample_moves=bool(args.sample_moves), opening_random_moves=args.v1_opening_random_moves, seed=seed)
    out = {"challenger_checkpoint": args.challenger_checkpoint, "previous_checkpoint": args.previous_checkpoint,
           "backend": "v1", "mcts_simulations": int(args.mcts_simulations), "seed": seed}
    if args.eval_games_vs_random > 0:
        out["vs_random"] = evaluate_checkpoint(args.challenger_checkpoint, None, num_games=args.eval_games_vs_random, **common)
    if args.eval_games_vs_previous > 0 and args.previous_checkpoint:
        out["vs_previous"] = evaluate_checkpoint(args.challenger_checkpoint, args.previous_checkpoint,
                                                 num_games=args.eval_games_vs_previous, **common)
    for key in ("vs_random", "vs_previous"):
        if key in out:
            p = out[key]
            print(f"[eval] {args.match_name or key}: W-L-D={p['wins']}-{p['losses']}-{p['draws']} ({p['total_games']} games), "
                  f"win={p['win_rate'] * 100:.2f}% loss={p['loss_rate'] * 100:.2f}% draw={p['draw_rate'] * 100:.2f}%", flush=True)
    if args.output_json:
        os.makedirs(os.path.dirname(args.output_json) or ".", exist_ok=True)
        with open(args.output_json, "w") as f:
            json.dump(out, f, indent=2)
    return 0


if __name__ == "__main__":
    sys.exit(main())
