"""Experiment: per-phase s_memtime stamps of the network kernel's residual block 2 (build_exp/lib_stamps.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "build_exp", "lib_stamps.so")
if not os.path.exists(LIB):                      # the production library is built without the stamps
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = [os.path.join(ROOT, "liuzhou_amd", "csrc", f) for f in ("lz_ops.hip", "lz_engine.hip", "lz_net.hip", "lz_train.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-fvisibility=hidden", "-DLZ_EXP_STAMPS", "-o", LIB] + src)
os.environ["LZ_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
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
