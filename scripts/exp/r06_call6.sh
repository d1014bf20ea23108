#!/bin/bash
# round 6, GPU call 6: the split tree step -- tree / selfplay / fullsize tests, then its A/B, then (if green) the profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tree.py tests/test_gpu_net.py tests/test_gpu_selfplay.py tests/test_gpu_fullsize.py tests/test_gpu_streams.py -x -q > gpurun_out/r06_gputest_4.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_4.log; tail -15 gpurun_out/r06_gputest_4.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python scripts/micro/tree_ab.py > gpurun_out/r06_tree_ab.jsonl 2> gpurun_out/r06_tree_ab.err
cat gpurun_out/r06_tree_ab.jsonl
for lib in default liuzhou_amd/_exp/liblz_X3_APF1.so default liuzhou_amd/_exp/liblz_X3_APF1.so; do
  if [ $lib = default ]; then unset LZ_HIP_LIB; else export LZ_HIP_LIB=$PWD/$lib; fi
  timeout -k 10 120 python scripts/micro/net_modes.py 2>/dev/null | grep fp16x3 | sed "s#^#{\"lib\": \"$lib\"} #" >> gpurun_out/r06_x3_prefetch_ab.jsonl || exit 1
done
unset LZ_HIP_LIB
cut -c1-330 gpurun_out/r06_x3_prefetch_ab.jsonl
