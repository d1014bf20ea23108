"""Experiment: cold vs hot (L2-resident) selection pass over 4096 searched trees (latency hypothesis of the tree kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import PortableTreeMCTS
from liuzhou_amd.mcts_gpu import GpuStateBatch
dev = torch.device("cuda:0")
torch.manual_seed(20260314)
net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
B = 4096
mcts = PortableTreeMCTS(net, B, 200, dev, use_graph=False)
state = GpuStateBatch.initial(dev, B)
out = mcts.search_batch(state, temperatures=torch.ones(B, device=dev))
e = mcts.engine
x = torch.zeros(64 << 20, device=dev)            # 256 MB: flush L2 / MALL
def timed(fn, n=1):
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e3 / n
print("path_len mean/max", e.buf["path_len"].float().mean().item(), e.buf["path_len"].max().item())
for rep in range(3):
    x.add_(1.0); torch.cuda.synchronize()
    cold = timed(e.select)
    hot = [timed(e.select) for _ in range(3)]
    print(f"select cold {cold:.1f} us, hot {hot[0]:.1f} {hot[1]:.1f} {hot[2]:.1f} us (path_len mean {e.buf['path_len'].float().mean().item():.2f})")
