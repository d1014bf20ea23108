#!/bin/bash
# round 6, GPU call 16: the final state -- smoke(), the whole GPU suite, the default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_final4.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_final4.log; tail -4 gpurun_out/r06_gputest_final4.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default_final2.json 2> gpurun_out/r06_bench_default_final2.err
echo "bench rc=$?"
tail -c 1000 gpurun_out/r06_bench_default_final2.json
