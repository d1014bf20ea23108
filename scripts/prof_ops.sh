#!/bin/bash
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}   # default: the repo this script lives in
OUT=$PWD/gpurun_out/prof_ops
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/scripts/bench_ops.py" > "$OUT/log.txt" 2>&1
grep -E "operator|encode|states_to|project|apply|pack" "$OUT/log.txt" | head -12
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/*/*kernel_stats.csv")[0])))
for r in rows:
    if any(k in r["Name"] for k in ("encode_actions", "model_input", "project_policy", "apply_moves", "pack_rows", "unpack_rows", "root_pack")):
        print("%-70s calls=%s avg_us=%.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
