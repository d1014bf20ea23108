"""Stress check of the two-stream search: 40 consecutive moves of 1024 games (graph replay, subtree reuse, re-seated
games), every move compared with a single-stream engine on the same states."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd import v0_core
from liuzhou_amd.mcts_gpu import GpuStateBatch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import DualStreamTreeMCTS, PortableTreeMCTS
dev = torch.device("cuda:0")
torch.manual_seed(20260314)
net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
B, sims = 1024, 24
kw = dict(add_dirichlet_noise=False, sample_moves=False, reuse_tree=True)
single = PortableTreeMCTS(net.variant(half_workgroups=True), B, sims, dev, **kw)
dual = DualStreamTreeMCTS(net, B, sims, dev, **kw)
st = GpuStateBatch.initial(dev, B)
plies = torch.zeros(B, dtype=torch.int64, device=dev); done = torch.zeros(B, dtype=torch.bool, device=dev)
reset = torch.zeros(B, dtype=torch.uint8, device=dev)
temps = torch.full((B,), 0.1, device=dev)
bad = 0
for mv in range(40):
    a = single.search_batch(st, temperatures=temps, reset=reset)
    b = dual.search_batch(st, temperatures=temps, reset=reset)
    same = torch.equal(a.chosen_action_indices, b.chosen_action_indices) and torch.equal(a.policy_dense, b.policy_dense) \
        and torch.equal(a.root_value, b.root_value)
    bad += 0 if same else 1
    reset.zero_()
    fin, _, _ = v0_core.self_play_step_inplace(*st.tensors(), plies, done, torch.arange(B, device=dev), a.chosen_action_codes,
                                               a.terminal_mask, a.chosen_valid_mask, 20 + (mv % 7), 2.0)
    if int(fin.numel()):                                   # re-seat finished games from the empty board
        fresh = GpuStateBatch.initial(dev, int(fin.numel()))
        for t, f in zip(st.tensors(), fresh.tensors()):
            t.index_copy_(0, fin, f)
        plies.index_fill_(0, fin, 0); done.index_fill_(0, fin, False); reset.index_fill_(0, fin, 1)
print("moves with a difference:", bad, "| dropped subtrees:", int(single.engine.reuse_dropped[0].item()),
      [int(p.engine.reuse_dropped[0].item()) for p in dual.parts])
assert bad == 0
