#!/usr/bin/env python3
"""Round-4 experiment: the product runner `self_play_v1_gpu` at the reference's workload (16 384 slots, 1 024 sims, 10x128)
with the table-driven bandit kernel and with the division kernel (LZ_ROOT_PUCT_DIV=1), same games, twice each."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu

dev = torch.device("cuda:0")
games = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
out = []
for rep in range(2):
    for div in ("0", "1"):
        os.environ["LZ_ROOT_PUCT_DIV"] = div
        torch.manual_seed(20260314)
        net = FusedNet(ChessNet(**MODEL_CONFIGS["b10c128"]).eval().to(dev))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        batch, st = self_play_v1_gpu(net, num_games=games * mult, mcts_simulations=sims, concurrent_games=games,
                                     temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
                                     exploration_weight=1.0, device=str(dev), add_dirichlet_noise=True, sample_moves=True,
                                     max_game_plies=512)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out.append({"division_kernel": div == "1", "rep": rep, "positions": int(batch.num_samples), "elapsed_s": round(dt, 2),
                    "positions_per_s": round(batch.num_samples / dt, 1), "timing_ms": st.step_timing_ms})
        print(json.dumps(out[-1]), flush=True)
        del batch, net
        torch.cuda.empty_cache()
