#!/bin/bash
# round 6, GPU call 17: the watch ignores searches whose launches the host spread out -- stream / worker tests, then the two
# worker legs and the C2 runner leg through bench.py's runner (does the worker still re-draw?)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_stream_worker.py tests/test_gpu_worker.py tests/test_gpu_selfplay.py -x -q > gpurun_out/r06_gputest_9.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_9.log; tail -4 gpurun_out/r06_gputest_9.log
[ $rc -eq 0 ] || exit 1
LZ_BENCH_C3_FULL=0 timeout -k 10 400 python bench.py --workload C2 --steps 50 --warmup 5 --also runner --no-cpu-baseline > gpurun_out/_w.json 2> gpurun_out/_w.err || { tail -5 gpurun_out/_w.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/_w.json").read().strip().splitlines()[-1])
r = d["also"]["runner"]
print(json.dumps({"C2": d["value"], "streams": d.get("streams"),
                  "legs": {k: [v.get("value"), v.get("stream_redraws"), (v.get("stream") or {}).get("redraws"), v.get("stream_overlap_pct")] for k, v in r.items()}}))
PY
