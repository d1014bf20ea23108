#!/bin/bash
# round 6, GPU call 7: the whole GPU suite on the final build, the round's rocprofv3 profiles, the default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_final.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_final.log; tail -6 gpurun_out/r06_gputest_final.log
[ $rc -eq 0 ] || exit 1
bash scripts/exp/prof_r06.sh || exit 1
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r06_bench_default.json
