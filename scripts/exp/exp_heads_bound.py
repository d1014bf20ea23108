"""Upper bound of taking the head phases out of the persistent network kernel: sustained evaluations/s with the pass cut
short after the trunk (LZ_NET_DEBUG_STOP=3) / after the head convs (=4) against the full pass (wrong results, timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
for name, N in (("b10c128", 16384), ("b6c64", 4096)):
    torch.manual_seed(20260314)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
    packed = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    packed[:, 0] = torch.randint(0, 1 << 36, (N,), device=dev) | (1 << 50)
    packed[:, 1] = torch.randint(0, 1 << 36, (N,), device=dev) & ~packed[:, 0] & ((1 << 36) - 1)
    for rep in range(2):
        for stop in ("0", "3", "4", "5"):
            os.environ["LZ_NET_DEBUG_STOP"] = stop
            for _ in range(3):
                f.forward_packed(packed)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 2.5:
                for _ in range(10):
                    f.forward_packed(packed)
                torch.cuda.synchronize(); n += 10
            dt = time.perf_counter() - t0
            print(f"{name} N={N} stop={stop}: {dt / n * 1e6:.1f} us/launch, {n * N / dt / 1e6:.3f} M evals/s", flush=True)
os.environ["LZ_NET_DEBUG_STOP"] = "0"
