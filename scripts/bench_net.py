import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
for name, Ns in (("b6c64", (4096, 16384, 65536, 262144)), ("b10c128", (2048, 16384, 65536))):
    torch.manual_seed(20260314)
    m = ChessNet(**MODEL_CONFIGS[name]).eval().to(dev)
    f = FusedNet(m)
    for N in Ns:
        x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
        for _ in range(2):
            f(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        s.record()
        for _ in range(reps):
            f(x, want_logits=False)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        tf = N * f.flops_per_eval / (ms * 1e-3) / 1e12
        print(f"{name} N={N}: {ms:.3f} ms  {N / ms * 1e3 / 1e6:.2f} M evals/s  {tf:.1f} TFLOP/s ({tf / 2500 * 100:.1f}% of 2.5 PF)", flush=True)
