#!/bin/bash
# round 6, GPU call 11: the fp32 argmax of the descent -- every tree / search test, then its same-box A/B at C2 and C3
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tree.py tests/test_gpu_selfplay.py tests/test_gpu_fullsize.py tests/test_gpu_mcts_core.py tests/test_gpu_net.py tests/test_gpu_worker.py -x -q > gpurun_out/r06_gputest_6.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_6.log; tail -8 gpurun_out/r06_gputest_6.log
[ $rc -eq 0 ] || exit 1
LZ_AB_VAR=LZ_TREE_F32SEL timeout -k 10 700 python scripts/micro/tree_ab.py > gpurun_out/r06_tree_f32sel_ab.jsonl 2> gpurun_out/r06_tree_f32sel_ab.err
cat gpurun_out/r06_tree_f32sel_ab.jsonl
