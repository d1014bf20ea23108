#!/usr/bin/env python3
"""Config C1 (BASELINE.json configs[0]): time the REAL reference `src/` pure-Python tree MCTS self-play in this
container and our CPU oracle on the same config, to give the "pure-Python src/" figure provenance.
4 games, 32 sims/move, tiny net (seed 7), CPU fp32, batch_K=1, T 1.0 -> 0.1 at ply 10, c=1, no noise, seeds 0.
Container-only (needs /root/reference).  Output: one JSON line."""
import json, os, random, statistics, sys, time
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
import numpy as np
import torch

torch.set_num_threads(int(os.environ.get("C1_THREADS", "8")))


def tiny():
    from src.neural_network import ChessNet
    torch.manual_seed(7)
    return ChessNet(trunk_channels=8, num_blocks=1, policy_channels=4, value_channels=4, value_mlp_channels=8).eval()


def run_reference(games=4, sims=32):
    from src.mcts import MCTS
    from src.game_state import GameState
    from src.move_generator import apply_move
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    model = tiny()
    positions = 0
    t0 = time.perf_counter()
    for _ in range(games):
        state = GameState()
        mcts = MCTS(model, num_simulations=sims, exploration_weight=1.0, temperature=1.0, device="cpu",
                    add_dirichlet_noise=False, virtual_loss_weight=0.0, batch_K=1)
        ply = 0
        while not state.is_game_over():
            mcts.temperature = 1.0 if ply < 10 else 0.1
            moves, policy = mcts.search(state)
            if not moves:
                break
            k = int(np.random.choice(len(moves), p=policy / policy.sum()))
            state = apply_move(state, moves[k], quiet=True)
            mcts.root = None
            positions += 1
            ply += 1
    dt = time.perf_counter() - t0
    return positions, dt


def run_oracle(games=4, sims=32):
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from oracle import selfplay_oracle as SO
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval()
    st = SO.self_play_tree(model, num_games=games, sims=sims)
    return st["num_positions"], st["elapsed_sec"]


if __name__ == "__main__":
    reps = int(os.environ.get("C1_REPS", "3"))
    ref = [run_reference() for _ in range(reps)]
    ours = [run_oracle() for _ in range(reps)]
    med = lambda xs: statistics.median(p / t for p, t in xs)
    print(json.dumps({"config": "C1: 4 games, 32 sims/move, tiny net, CPU fp32, batch_K=1", "threads": torch.get_num_threads(),
                      "reference_src_positions_per_sec": round(med(ref), 2), "reference_runs": ref,
                      "oracle_port_positions_per_sec": round(med(ours), 2), "oracle_runs": ours,
                      "ratio_oracle_over_reference": round(med(ours) / med(ref), 2)}))
