"""Experiment: shader clock held inside the network kernel (LZ_NET_DEBUG_STOP=99 writes s_memtime / wall-clock deltas)."""
import os, sys
os.environ["LZ_NET_DEBUG_STOP"] = "99"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
for name, N in (("b6c64", 4096), ("b6c64", 65536), ("b10c128", 16384)):
    torch.manual_seed(20260314)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
    x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
    for _ in range(30):
        f(x, want_logits=False)
    torch.cuda.synchronize()
    v = f.last_value[:2].cpu().tolist()
    print(f"{name} N={N}: memtime ticks {v[0]:.0f}, wall ticks {v[1]:.0f} (100 MHz) -> {v[0] / max(v[1], 1) * 100:.0f} MHz, "
          f"kernel span {v[1] / 100:.1f} us", flush=True)
