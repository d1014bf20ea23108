#!/bin/bash
# round 6, GPU call 3: the GPU suite on the new build, then C3 with two engines on two streams (LZ_DUAL_128) against one
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_2.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_2.log; tail -15 gpurun_out/r06_gputest_2.log
[ $rc -eq 0 ] || exit 1
for mode in single dual single dual; do
  if [ $mode = dual ]; then export LZ_DUAL_128=1; D=1; else unset LZ_DUAL_128; D=0; fi
  timeout -k 10 200 python bench.py --workload C3 --steps 10 --warmup 3 --also none --no-cpu-baseline --dual-stream $D \
      > gpurun_out/_b.json 2> gpurun_out/_b.err || { tail -5 gpurun_out/_b.err; exit 1; }
  python - "$mode" <<'PY' >> gpurun_out/r06_c3_dual128.jsonl
import json, sys
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(json.dumps({"mode": sys.argv[1], "dual_stream": d["config"]["dual_stream"], "value": d["value"], "ms": d["ms_per_step"],
                  "frac": d["roofline"]["frac"], "streams": d.get("streams"), "clocks": {k: d["clocks"].get(k) for k in ("power_w_mean", "sclk_mhz_mean")}}))
PY
done
cat gpurun_out/r06_c3_dual128.jsonl
