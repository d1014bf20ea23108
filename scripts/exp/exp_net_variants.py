"""Sustained evaluations/s of the network kernel for several builds of the HIP library (LZ_HIP_LIB), one child process
per build: the regular build and the timing-experiment builds that drop one kind of operand traffic
(-DLZ_EXP_NO_BRELOAD: no LDS activation-operand reloads, -DLZ_EXP_NO_ALOAD: no weight-fragment loads; wrong results).
Under the package power limit this bounds what restructuring the operand delivery could gain."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
for name, N in (("b10c128", 16384), ("b6c64", 65536)):
    torch.manual_seed(20260314)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
    packed = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    packed[:, 0] = torch.randint(0, 1 << 36, (N,), device=dev) | (1 << 50)
    packed[:, 1] = torch.randint(0, 1 << 36, (N,), device=dev) & ~packed[:, 0] & ((1 << 36) - 1)
    for _ in range(3):
        f.forward_packed(packed)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(10):
            f.forward_packed(packed)
        torch.cuda.synchronize(); n += 10
    dt = time.perf_counter() - t0
    smi = os.popen("rocm-smi --showpower --showclocks --csv | tail -2 | head -1").read().strip()
    print(f"  {name} N={N}: {n * N / dt / 1e6:.3f} M evals/s = {n * N / dt * f.flops_per_eval / 1e12:.0f} TFLOP/s (idle-after: {smi[:90]})", flush=True)
''' % ROOT
for tag, lib in (("regular", None), ("NO_BRELOAD", "liuzhou_amd/_exp/liblz_NO_BRELOAD.so"), ("NO_ALOAD", "liuzhou_amd/_exp/liblz_NO_ALOAD.so")):
    env = dict(os.environ)
    if lib:
        env["LZ_HIP_LIB"] = os.path.join(ROOT, lib)
    print(tag, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env)
