"""HBM roofline of the byte/integer operators of the path (algorithmic bytes per unit):
kernel-only durations from rocprofv3 (run under scripts/prof_ops.sh) or HIP-event timings when run directly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from liuzhou_amd import v0_core
from liuzhou_amd.mcts_gpu import GpuStateBatch
from liuzhou_amd.steady_state import SteadyStateRootSelfPlay
from liuzhou_amd.mcts_gpu import V1RootMCTSConfig
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.trajectory_codec import pack_batch, unpack_records
from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch

dev = torch.device("cuda:0")
B = 65536
torch.manual_seed(0)
# mid-game states: a steady-state population rolled forward with the root search
pop = SteadyStateRootSelfPlay(FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev)), B,
                              V1RootMCTSConfig(num_simulations=2), dev, seed=1)
pop.preroll(90)
st = pop.states
t = st.tensors()


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


rows = []
mask, meta = v0_core.encode_actions_fast(*t[:10], 36, 144, 36, 4)
rows.append(("encode_actions_fast", B, 180 + 220 + 3520, timed(lambda: v0_core.encode_actions_fast(*t[:10], 36, 144, 36, 4))))
planes = v0_core.states_to_model_input(*t[:5])
rows.append(("states_to_model_input", B, 180 + 1584, timed(lambda: v0_core.states_to_model_input(*t[:5]))))
lp = [torch.log_softmax(torch.randn(B, 36, device=dev), 1) for _ in range(3)]
rows.append(("project_policy_logits_fast", B, 432 + 220 + 1760, timed(lambda: v0_core.project_policy_logits_fast(*lp, mask, 36, 144, 36, 4))))
probs, _ = v0_core.project_policy_logits_fast(*lp, mask, 36, 144, 36, 4)
pack = v0_core.root_pack_sparse_actions(mask, probs, meta)
codes_all, parents_all = pack[8], pack[9]
N = int(codes_all.shape[0])
rows.append(("batch_apply_moves (children of all states)", N, 24 + 180 + 180,
             timed(lambda: v0_core.batch_apply_moves(*t, codes_all, parents_all))))
val = torch.zeros(B, device=dev); soft = torch.zeros(B, device=dev)
batch = TensorSelfPlayBatch(planes, mask, probs, val, soft)
rec = pack_batch(batch)
rows.append(("pack_trajectory_rows", B, 2692 + 360, timed(lambda: pack_batch(batch))))
rows.append(("unpack_trajectory_rows", B, 2692 + 360, timed(lambda: unpack_records(rec))))
print(f"{'operator':46s} {'units':>9s} {'B/unit':>7s} {'us':>9s} {'GB/s':>8s} {'% of 8 TB/s':>11s}")
for name, n, bpu, us in rows:
    gbs = n * bpu / (us * 1e-6) / 1e9
    print(f"{name:46s} {n:9d} {bpu:7d} {us:9.1f} {gbs:8.0f} {gbs / 8000 * 100:10.1f}%")
print("(HIP-event time of the whole Python call: includes output allocation and argument marshalling; "
      "kernel-only durations are in profiles/r01_ops_kernel_stats.md)")
