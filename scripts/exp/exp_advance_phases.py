#!/usr/bin/env python3
"""Where does lz_tree_advance spend its time?  C2 population (4 096 games, 200 sims, two engines), 40 steps; per launch the
slowest game's phase durations (100 MHz ticks -> us) and its node counts."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd import _lib as L
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay

dev = torch.device("cuda:0")
games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(20260314)
net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
pop = SteadyStateTreeSelfPlay(net, games, sims=sims, device=dev, seed=9973, reuse_tree=True, dual_stream=False, arena_rows=games * 200)
pop.mcts.use_graph = False
pop.preroll(120)
pop.prepare()
for _ in range(30):
    pop.step()
ticks = torch.zeros((games, 8), dtype=torch.int64, device=dev)
L.check(L.lib().lz_debug_advance_ticks(L.ptr(ticks)), "debug_advance_ticks")
rows = []
for step in range(10):
    ticks.zero_()
    pop.step()
    torch.cuda.synchronize(dev)
    t = ticks.cpu()
    kept = t[:, 3] > 0
    if not bool(kept.any()):
        continue
    d1, d2, d3 = (t[:, 1] - t[:, 0])[kept], (t[:, 2] - t[:, 1])[kept], (t[:, 3] - t[:, 2])[kept]
    tot = (t[:, 3] - t[:, 0])[kept]
    i = int(tot.argmax())
    span = (t[:, 3][kept].max() - t[:, 0][t[:, 0] > 0].min()).item() / 100.0
    print(f"step {step}: games keeping a subtree {int(kept.sum())}, kernel span {span:.0f} us; slowest game: marks {d1[i] / 100:.0f} us, "
          f"nodes {d2[i] / 100:.0f} us, runs {d3[i] / 100:.0f} us, nodes {int(t[:, 4][kept][i])} -> kept {int(t[:, 5][kept][i])}; "
          f"mean game: {tot.float().mean() / 100:.0f} us, nodes {t[:, 4][kept].float().mean():.0f} -> {t[:, 5][kept].float().mean():.0f}; "
          f"max nodes {int(t[:, 4].max())}", flush=True)
L.lib().lz_debug_advance_ticks(None)
