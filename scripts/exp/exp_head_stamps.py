"""Experiment: s_memtime stamps of every wave of one workgroup at the head sub-steps of the network kernel
(build_exp/lib_head_stamps.so, -DLZ_EXP_HEAD_STAMPS)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "build_exp", "lib_head_stamps.so")
if not os.path.exists(LIB):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = [os.path.join(ROOT, "liuzhou_amd", "csrc", f) for f in ("lz_ops.hip", "lz_engine.hip", "lz_net.hip", "lz_net_f32.hip", "lz_train.hip", "lz_search.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-fvisibility=hidden", "-DLZ_EXP_HEAD_STAMPS", "-o", LIB] + src)
os.environ["LZ_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
names = ["trunk store + head convs", "store p", "gpool p", "gpool-linear", "out convs", "log-softmax", "store v", "gpool v", "fc1", "fc2",
         "expectation"]
for name, N, half in (("b6c64", 2048, True), ("b6c64", 4096, False), ("b10c128", 16384, False)):
    torch.manual_seed(20260314)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev)).variant(half_workgroups=half)
    x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
    for _ in range(20):
        out = f(x, want_logits=True)
    torch.cuda.synchronize()
    nw = 4 if half else 8
    raw = out[3].view(-1)[:nw * 24].view(nw, 24).cpu()
    st = raw[:, :12]
    print(name, N, "half workgroups" if half else "")
    front = ["net_setup", "first barrier", "staging loop", "barrier", "stem", "residual blocks"]
    for w in range(0, nw, max(1, nw // 4)):
        d = (st[w, 1:] - st[w, :-1]).tolist()
        print(f"  wave {w}: " + "  ".join(f"{n} {int(v)}" for n, v in zip(front, raw[w, 12:18].tolist())))
        print(f"          " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, d)) + f"  | heads {int(st[w, 11])}")
        print(f"          record load before the barrier {int(raw[w, 18])}  first barrier alone {int(raw[w, 19])}  entry clock (low 24 bits) {int(raw[w, 20])}")
