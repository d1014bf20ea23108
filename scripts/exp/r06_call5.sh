#!/bin/bash
# round 6, GPU call 5: tree + network tests on the new build, split-conv prefetch A/B, then the round's rocprofv3 profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tree.py tests/test_gpu_net.py -x -q > gpurun_out/r06_gputest_3.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_3.log; tail -12 gpurun_out/r06_gputest_3.log
[ $rc -eq 0 ] || exit 1
for lib in default liuzhou_amd/_exp/liblz_X3_APF1.so default liuzhou_amd/_exp/liblz_X3_APF1.so; do
  if [ $lib = default ]; then unset LZ_HIP_LIB; else export LZ_HIP_LIB=$PWD/$lib; fi
  timeout -k 10 120 python scripts/micro/net_modes.py 2>/dev/null | grep fp16x3 | sed "s#^#{\"lib\": \"$lib\"} #" >> gpurun_out/r06_x3_prefetch_ab.jsonl || exit 1
done
unset LZ_HIP_LIB
cat gpurun_out/r06_x3_prefetch_ab.jsonl | cut -c1-330
bash scripts/exp/prof_r06.sh
