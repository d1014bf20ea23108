#!/bin/bash
# round 6, GPU call 4: the split-operand (fp16x3) mode -- its tests, its speed next to fp16 / fp32, its C2 bench leg
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_net.py -x -q > gpurun_out/r06_gputest_net.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_net.log; tail -25 gpurun_out/r06_gputest_net.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python scripts/micro/net_modes.py > gpurun_out/r06_net_modes.jsonl 2> gpurun_out/r06_net_modes.err || { tail -5 gpurun_out/r06_net_modes.err; exit 1; }
cat gpurun_out/r06_net_modes.jsonl
timeout -k 10 300 python bench.py --workload C2 --steps 20 --warmup 3 --also C2_fp32,C2_fp16x3 --no-cpu-baseline > gpurun_out/_b.json 2> gpurun_out/_b.err || { tail -5 gpurun_out/_b.err; exit 1; }
python - <<'PY' | tee gpurun_out/r06_c2_parity_modes.json
import json
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(json.dumps({k: {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"), "frac": (v.get("roofline") or {}).get("frac"),
                      "peak": (v.get("roofline") or {}).get("peak"), "dtype": v.get("dtype")}
                  for k, v in [("C2_fp16", d)] + list(d.get("also", {}).items())}))
PY
