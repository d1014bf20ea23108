import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
for C in (64, 128):
    for nb in (0, 1, 2, 6, 10):
        torch.manual_seed(1)
        m = ChessNet(trunk_channels=C, num_blocks=nb).eval().to(dev)
        f = FusedNet(m)
        N = 4096 * (1 if C == 64 else 1) // (1 if C == 64 else 2)
        x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
        for _ in range(3): f(x, want_logits=False)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f(x, want_logits=False)
        e.record(); torch.cuda.synchronize()
        print(f"C={C} blocks={nb} N={N}: {s.elapsed_time(e) / 10 * 1000:.1f} us per launch (1 pass per CU)", flush=True)
