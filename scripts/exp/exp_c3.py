"""Experiments on the 10x128 network kernel at the C3 launch shape (16 384 evaluations per launch):
  * evaluations/s against the number of persistent workgroups (= CUs used): under the package power limit fewer CUs
    may cost less than their share (the clock rises), which decides whether CUs can be set aside for the tree kernel;
  * the same with every trunk layer reading one block's weights (a library built with -DLZ_EXP_SAME_LAYER, wrong
    results, never the shipped build -- `LZ_EXP_SAME_LAYER=1` in the environment only labels the run): the weight set
    then fits the 4 MB XCD L2, which bounds what the L2 misses on the real 5.9 MB set cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "b10c128"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
torch.manual_seed(20260314)
model = ChessNet(**MODEL_CONFIGS[name]).eval().to(dev)
packed = torch.zeros((N, 4), dtype=torch.int64, device=dev)
packed[:, 0] = torch.randint(0, 1 << 36, (N,), device=dev) | (1 << 50)
packed[:, 1] = torch.randint(0, 1 << 36, (N,), device=dev) & ~packed[:, 0] & ((1 << 36) - 1)


def rate(f, seconds=2.5):
    for _ in range(3):
        f.forward_packed(packed)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            f.forward_packed(packed)
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    return n * N / dt, dt / n * 1e6


mode = sys.argv[3] if len(sys.argv) > 3 else "cus"
if mode == "wide":          # 8-wave shape (2 channel tiles per wave) vs 4-wave shape (4 channel tiles per wave), alternating
    for rep in range(3):
        for wide in (False, True):
            f = FusedNet(model, wide_tiles=wide)
            r, us = rate(f, 3.0)
            print(f"{name} N={N} wide_tiles={wide}: {r / 1e6:.3f} M evals/s, {us:.1f} us/launch, "
                  f"{r * f.flops_per_eval / 1e12:.0f} TFLOP/s", flush=True)
    sys.exit(0)
same = os.environ.get("LZ_EXP_SAME_LAYER", "0")      # label only: the behaviour is a compile-time flag of the library
for _ in (0,):
    for blocks in (256, 248, 240, 224, 192, 128):
        f = FusedNet(model, max_blocks=blocks)
        r, us = rate(f)
        print(f"{name} N={N} same_layer={same} workgroups={blocks}: {r / 1e6:.3f} M evals/s, {us:.1f} us/launch, "
              f"{r * f.flops_per_eval / 1e12:.0f} TFLOP/s", flush=True)
