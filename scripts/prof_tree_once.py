"""Single-workload driver for profiling the tree kernels (tree_expand_select_kernel, tree_advance_kernel): a few searched
moves of a steady-state population with ONE engine on one stream and direct launches (no graph), so that every kernel
of the search is its own record in a rocprofv3 trace.  usage: prof_tree_once.py <model> <games> <sims> <moves> [compact]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
model = sys.argv[1] if len(sys.argv) > 1 else "b6c64"
games = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 200
moves = int(sys.argv[4]) if len(sys.argv) > 4 else 3
os.environ["LZ_TREE_COMPACT"] = sys.argv[5] if len(sys.argv) > 5 else "0"
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay
dev = torch.device("cuda:0")
torch.manual_seed(20260314)
net = FusedNet(ChessNet(**MODEL_CONFIGS[model]).eval().to(dev))
pop = SteadyStateTreeSelfPlay(net, games, sims=sims, device=dev, seed=9973, reuse_tree=True, reuse_factor=8.0,
                              dual_stream=False, arena_rows=games * (moves + 8))
pop.mcts.use_graph = False
pop.preroll(120)
for _ in range(moves):
    pop.step()
torch.cuda.synchronize()
print("searched", moves, "moves of", games, "games,", sims, "sims,", model)
