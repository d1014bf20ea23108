#!/bin/bash
# round 6, GPU call 18: the committed tree as the driver will see it -- smoke(), the whole GPU suite, the default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke2.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06_smoke2.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_final5.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_final5.log; tail -4 gpurun_out/r06_gputest_final5.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r06_bench_default_final3.json 2> gpurun_out/r06_bench_default_final3.err
echo "bench rc=$?"
tail -c 1200 gpurun_out/r06_bench_default_final3.json
