#!/usr/bin/env python3
"""Provenance of the CPU baseline at C2's ARITHMETIC (the reference-to-port ratio had only been
measured on C1's tiny net): the REAL reference `src/` pure-Python tree MCTS (batch_K = 1, no noise) and our CPU oracle,
both with the 6-block / 64-channel net at 200 simulations per move, a bounded number of plies of one game each, in this
container.  Container-only (needs /root/reference).  Output: one JSON line.

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_c2.py [plies]"""
import json, os, random, sys, time
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
import numpy as np
import torch

THREADS = int(os.environ.get("C2_THREADS", "8"))
torch.set_num_threads(THREADS)
PLIES = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SIMS = 200


def run_reference():
    from src.neural_network import ChessNet
    from src.mcts import MCTS
    from src.game_state import GameState
    from src.move_generator import apply_move
    random.seed(0); np.random.seed(0); torch.manual_seed(20260314)
    model = ChessNet(trunk_channels=64, num_blocks=6, policy_channels=64, value_channels=64, value_mlp_channels=128).eval()
    state = GameState()
    mcts = MCTS(model, num_simulations=SIMS, exploration_weight=1.0, temperature=1.0, device="cpu",
                add_dirichlet_noise=False, virtual_loss_weight=0.0, batch_K=1)
    t0 = time.perf_counter()
    n = 0
    while not state.is_game_over() and n < PLIES:
        moves, policy = mcts.search(state)
        if not moves:
            break
        k = int(np.random.choice(len(moves), p=policy / policy.sum()))
        state = apply_move(state, moves[k], quiet=True)
        mcts.root = None
        n += 1
    return n, time.perf_counter() - t0


def run_oracle():
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from oracle import selfplay_oracle as SO
    torch.manual_seed(20260314)
    model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval()
    st = SO.self_play_tree(model, num_games=1, sims=SIMS, max_game_plies=PLIES, reuse_tree=False)
    return st["num_positions"], st["elapsed_sec"]


if __name__ == "__main__":
    ref = run_reference()
    ours = run_oracle()
    r, o = ref[0] / ref[1], ours[0] / ours[1]
    print(json.dumps({"config": f"C2 arithmetic: 1 game, first {PLIES} plies, {SIMS} sims/move, 6x64 net, CPU fp32, batch_K=1, no reuse",
                      "threads": THREADS, "reference_src_positions_per_sec": round(r, 3), "reference_run": ref,
                      "oracle_port_positions_per_sec": round(o, 3), "oracle_run": ours,
                      "ratio_oracle_over_reference": round(o / r, 2)}))
