#!/bin/bash
# round 6, GPU call 14: (1) tests of the templated advance kernel / large arenas; (2) what the overlap watch SEES in the
# reproducibly serialised C2 runner leg of a priority pair (LZ_WATCH_DEBUG)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tree.py tests/test_gpu_mcts_core.py tests/test_gpu_fullsize.py tests/test_gpu_selfplay.py tests/test_gpu_worker.py -x -q > gpurun_out/r06_gputest_7.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_7.log; tail -8 gpurun_out/r06_gputest_7.log
[ $rc -eq 0 ] || exit 1
LZ_WATCH_DEBUG=1 LZ_STREAM_PAIR=priority LZ_BENCH_C3_FULL=0 timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/_p.json 2> gpurun_out/r06_watch_debug.err
grep "lz watch" gpurun_out/r06_watch_debug.err | tail -40
python - <<'PY'
import json
d = json.loads(open("gpurun_out/_p.json").read().strip().splitlines()[-1])
r = d["also"]["runner"]
print({k: r[k]["value"] for k in r})
PY
