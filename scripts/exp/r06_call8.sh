#!/bin/bash
# round 6, GPU call 8: stream tests after the watch change, then the default bench of the final build
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_streams.py tests/test_gpu_selfplay.py -x -q > gpurun_out/r06_gputest_5.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_5.log; tail -4 gpurun_out/r06_gputest_5.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
echo "bench rc=$?"
tail -c 1300 gpurun_out/r06_bench_default.json
