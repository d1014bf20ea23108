"""liuzhou_amd -- MI355X-native self-play hot path for Liuzhou Chess.

Package layout (only what the path needs):
  csrc/            hand-written gfx950 kernels + C ABI (include/liuzhou_hip.h)
  v0_core.py       the reference's `v0_core` operator surface over the C ABI
  mcts_gpu.py      GpuStateBatch / V1RootMCTS (root-PUCT search)
  tree_engine.py   device-resident full-tree PUCT engine
  self_play_*.py   wave loop, trajectory arena, worker / shard writer
  net.py           policy + bucketed-value ResNet (same state_dict as the reference)
"""
__version__ = "0.1.0"
