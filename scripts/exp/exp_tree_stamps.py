"""Experiment: where a wave of `tree_expand_select_kernel` spends its time (build_exp/lib_tree_stamps.so,
-DLZ_EXP_TREE_STAMPS): one game's wave adds its clock to a global accumulator at every stamp (waiting for outstanding
memory operations first, so a wait is charged to the step that caused it); averages per simulation over a steady-state run.
usage: python scripts/exp/exp_tree_stamps.py [C2|C3]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "build_exp", "lib_tree_stamps.so")
if not os.path.exists(LIB):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = [os.path.join(ROOT, "liuzhou_amd", "csrc", f) for f in ("lz_ops.hip", "lz_engine.hip", "lz_net.hip", "lz_net_f32.hip", "lz_train.hip", "lz_search.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-fvisibility=hidden", "-DLZ_EXP_TREE_STAMPS", "-o", LIB] + src)
os.environ["LZ_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import torch
from liuzhou_amd import _lib as L
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay

which = sys.argv[1] if len(sys.argv) > 1 else "C2"
games, sims, net, dual, steps = (4096, 200, "b6c64", True, 12) if which == "C2" else (16384, 800, "b10c128", False, 2)
dev = torch.device("cuda:0")
torch.manual_seed(20260314)
model = ChessNet(**MODEL_CONFIGS[net]).eval().to(dev)
pop = SteadyStateTreeSelfPlay(model, games, sims=sims, device=dev, reuse_tree=True, dual_stream=dual, arena_rows=games * 40)
pop.preroll(120)
pop.prepare()
for _ in range(3):
    pop.step()
torch.cuda.synchronize()
lib = L.lib()
buf = (C.c_ulonglong * 32)()
assert lib.lz_exp_tree_stamps(buf, 1) == 0
for _ in range(steps):
    pop.step()
torch.cuda.synchronize()
assert lib.lz_exp_tree_stamps(buf, 0) == 0
s = list(buf)
n = max(1, s[19])
names = ["kernel entry", "independent loads", "legal set + head gather + softmax", "compaction + renormalisation sum", "allocation",
         "child states / terminal tests / edge records", "backup issued", "fence (+ root reload)", "descent", "leaf state"]
print(which, "kernel runs of the stamped waves:", n, " levels per descent:", round(s[18] / n, 2))
print("  mean clock since kernel entry at each stamp (a stamp waits for outstanding memory operations first), and how often it ran:")
prev = 0.0
for k, name in enumerate(names):
    cnt = s[20 + k]
    at = s[k] / max(1, cnt)
    print(f"  {name:48s} at {at:9.0f}  (+{at - prev:7.0f})  x{cnt}")
    prev = at
print(f"  inside the descent, per level: wait for the edge run {s[16] / max(1, s[18]):.0f}, scores + maximum + broadcast {s[17] / max(1, s[18]):.0f}")
