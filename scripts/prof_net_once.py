"""Single-kernel workload for PMC passes: the fused network kernel at the bench batch (4096 evals, 6x64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "b6c64"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
half = len(sys.argv) > 3 and sys.argv[3] == "half"          # 4-wave / 8-sample workgroups (dual-stream search)
torch.manual_seed(20260314)
f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev), half_workgroups=half)
x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
for _ in range(20):
    f(x, want_logits=False)
torch.cuda.synchronize()
