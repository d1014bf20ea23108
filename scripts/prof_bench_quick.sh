#!/bin/bash
# kernel-stats of a short default bench run: prints the top kernels (avg us)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}   # default: the repo this script lives in
OUT=$PWD/gpurun_out/prof_quick
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/*/*kernel_stats.csv")[0])))
for r in rows[:int("${TOPN:-6}")]:
    print("%-60s calls=%s avg_us=%.2f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
