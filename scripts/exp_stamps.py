"""Experiment: per-phase s_memtime stamps of the network kernel's residual block 2 (build_exp/lib_stamps.so)."""
import os, sys
os.environ["LZ_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_exp", "lib_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
names = ["->barrierA", "store1", "->barrierB", "conv1", "->barrierA2", "store2", "->barrierB2", "conv2"]
for name, N in (("b6c64", 4096), ("b6c64", 65536), ("b10c128", 16384)):
    torch.manual_seed(20260314)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
    x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
    for _ in range(20):
        out = f(x, want_logits=True)
    torch.cuda.synchronize()
    st = out[3].view(-1)[:128].view(8, 16)[:, :9].cpu()
    print(name, N)
    for w in (0, 3, 4, 7):
        d = (st[w, 1:] - st[w, :-1]).tolist()
        print(f"  wave {w}: " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, d)) + f"  | total {int(st[w, 8])} cycles")
