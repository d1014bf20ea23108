"""How many simulations of a steady-state population end in a leaf that needs the network?  (trace of the expand kernel:
kind 1 = evaluate, 2 = terminal leaf, 0 = inactive game, 3 = kept root.)  Decides whether compacting terminal leaves
out of the network batch would pay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay

dev = torch.device("cuda:0")
games, sims, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.manual_seed(20260314)
net = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
pop = SteadyStateTreeSelfPlay(net, games, sims=sims, device=dev, reuse_tree=True, dual_stream=False,
                              arena_rows=games * 64)
pop.mcts.engine.enable_trace()
pop.preroll(120)
pop.prepare()
tot = {0: 0, 1: 0, 2: 0, 3: 0}
depth_sum = 0.0
for step in range(int(sys.argv[4]) if len(sys.argv) > 4 else 12):
    pop.step()
    torch.cuda.synchronize()
    k = pop.mcts.engine.trace["trace_kind"]
    for v in tot:
        tot[v] += int((k == v).sum())
    depth_sum += float(pop.mcts.engine.buf["path_len"].float().mean())
n = sum(tot.values())
print(f"{name} {games} games x {sims} sims: evaluate {tot[1] / n:.4f}, terminal {tot[2] / n:.4f}, inactive {tot[0] / n:.4f}, "
      f"kept roots {tot[3] / n:.5f}; mean depth of the last descent {depth_sum / (step + 1):.2f}")
plies = pop.pop.plies.float()
print(f"plies: mean {plies.mean():.1f} max {plies.max():.0f}")
