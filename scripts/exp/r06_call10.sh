#!/bin/bash
# round 6, GPU call 10: the split tree step at C3's launch shape (16 384 games), same box A/B
set -o pipefail
mkdir -p gpurun_out
for m in 8192 16384 8192 16384; do
  LZ_TREE_SPLIT_MAX=$m timeout -k 10 200 python bench.py --workload C3 --steps 8 --warmup 3 --also none --no-cpu-baseline > gpurun_out/_b.json 2> gpurun_out/_b.err || { tail -5 gpurun_out/_b.err; exit 1; }
  python - "$m" <<'PY' >> gpurun_out/r06_c3_split.jsonl
import json, sys
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
sec = (d["roofline"].get("secondary") or {}).get("tree_expand_select_kernel", {})
print(json.dumps({"split_max": int(sys.argv[1]), "value": d["value"], "ms": d["ms_per_step"], "tree_us": (sec.get("as_scheduled") or {}).get("avg_launch_us"),
                  "sclk": d["clocks"].get("sclk_mhz_mean")}))
PY
done
cat gpurun_out/r06_c3_split.jsonl
