"""Single-workload driver for profiling the persistent search kernel (csrc/lz_search.hip): a few searched moves of the
C2 population (4 096 games, 200 simulations, 6x64) with LZ_TREE_PERSISTENT=1, direct launches (no graph)."""
import os, sys
os.environ["LZ_TREE_PERSISTENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay
dev = torch.device("cuda:0")
games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 200
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.manual_seed(20260314)
net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
pop = SteadyStateTreeSelfPlay(net, games, sims=sims, device=dev, seed=9973, reuse_tree=True, reuse_factor=8.0,
                              dual_stream=True, arena_rows=games * (moves + 8))
pop.mcts.use_graph = False
pop.preroll(120)
pop.prepare()
for _ in range(moves):
    pop.step()
torch.cuda.synchronize()
print("searched", moves, "moves; persistent:", pop.mcts.engine.persistent_ok(net))
