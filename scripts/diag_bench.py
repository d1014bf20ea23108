import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.mcts_gpu import V1RootMCTSConfig
from liuzhou_amd.steady_state import SteadyStateRootSelfPlay
LOG = open("gpurun_out/diag.log", "a")
def log(*a):
    print(*a, flush=True); print(*a, file=LOG, flush=True)
dev = torch.device("cuda:0")
games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(20260314)
model = ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev)
from liuzhou_amd.net_hip import FusedNet
model = FusedNet(model)
cfg = V1RootMCTSConfig(num_simulations=200, autocast_dtype=sys.argv[2] if len(sys.argv) > 2 else "float16")
t = time.time(); pop = SteadyStateRootSelfPlay(model, games, cfg, dev); torch.cuda.synchronize(); log("init", time.time() - t)
t = time.time(); pop.preroll(120); torch.cuda.synchronize(); log("preroll", time.time() - t)
for i in range(8):
    t = time.time(); pop.step(); torch.cuda.synchronize(); log("step", i, round(time.time() - t, 4), "evals", pop.mcts._leaf_evals)
