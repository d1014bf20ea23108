"""Single-kernel workload for PMC passes: the fused network kernel at a bench launch shape (model, batch, full|half)."""
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
# the launch shape of the search loop: 32-byte packed states in, three head rows + the value out
packed = torch.zeros((N, 4), dtype=torch.int64, device=dev)
packed[:, 0] = torch.randint(0, 1 << 36, (N,), device=dev) | (1 << 50)          # random black stones, phase 1
packed[:, 1] = torch.randint(0, 1 << 36, (N,), device=dev) & ~packed[:, 0] & ((1 << 36) - 1)
for _ in range(20):
    f.forward_packed(packed)
torch.cuda.synchronize()
