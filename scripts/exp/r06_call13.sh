#!/bin/bash
# round 6, GPU call 13: does the serialised C2 runner leg come back with a priority pair, and does the extended watch repair it?
# Two full default invocations with LZ_STREAM_PAIR=priority (the setting of the run in which it happened).
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
  LZ_STREAM_PAIR=priority LZ_BENCH_C3_FULL=0 timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/_p.json 2> gpurun_out/_p.err || { tail -5 gpurun_out/_p.err; exit 1; }
  python - "$i" <<'PY' >> gpurun_out/r06_priority_pair_repeats.jsonl
import json, sys
d = json.loads(open("gpurun_out/_p.json").read().strip().splitlines()[-1])
r = d["also"]["runner"]
print(json.dumps({"run": int(sys.argv[1]), "pair": (d["also"]["C2"].get("streams") or {}), "C2": d["also"]["C2"]["value"],
                  "runner_C2": [r["self_play_tree_gpu@C2"]["value"], r["self_play_tree_gpu@C2#warm"]["value"]],
                  "runner_C2_redraws": [r["self_play_tree_gpu@C2"].get("stream_redraws"), r["self_play_tree_gpu@C2#warm"].get("stream_redraws")],
                  "worker_C2": r["run_self_play_worker@C2_tree"]["value"], "worker_C2_redraws": (r["run_self_play_worker@C2_tree"].get("stream") or {}).get("redraws")}))
PY
done
cat gpurun_out/r06_priority_pair_repeats.jsonl
