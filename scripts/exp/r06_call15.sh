#!/bin/bash
# round 6, GPU call 15: the reference-based overlap watch -- stream / selfplay tests, then the default sequence with the
# priority pair (whose C2 runner leg ran at 31 ms per ply every time): is it noticed and repaired now?  Then the default pair.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_selfplay.py tests/test_gpu_tree.py tests/test_gpu_mcts_core.py -x -q > gpurun_out/r06_gputest_8.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_8.log; tail -6 gpurun_out/r06_gputest_8.log
[ $rc -eq 0 ] || exit 1
for pair in priority probe; do
  LZ_STREAM_PAIR=$pair LZ_BENCH_C3_FULL=0 timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/_p.json 2> gpurun_out/_p.err || { tail -5 gpurun_out/_p.err; exit 1; }
  python - "$pair" <<'PY' >> gpurun_out/r06_watch_reference.jsonl
import json, sys
d = json.loads(open("gpurun_out/_p.json").read().strip().splitlines()[-1])
r = d["also"]["runner"]
print(json.dumps({"pair_asked": sys.argv[1], "C2": d["also"]["C2"]["value"], "C2_streams": d["also"]["C2"].get("streams"),
                  "runner_C2": [r[k]["value"] for k in ("self_play_tree_gpu@C2", "self_play_tree_gpu@C2#warm")],
                  "runner_C2_redraws": [r[k].get("stream_redraws") for k in ("self_play_tree_gpu@C2", "self_play_tree_gpu@C2#warm")],
                  "runner_C2_overlap_pct": [r[k].get("stream_overlap_pct") for k in ("self_play_tree_gpu@C2", "self_play_tree_gpu@C2#warm")],
                  "worker_C2": r["run_self_play_worker@C2_tree"]["value"], "headline": d["value"]}))
PY
done
cat gpurun_out/r06_watch_reference.jsonl
