#!/bin/bash
# Profiles behind bench.py's roofline numbers (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench workload (tree search, C2)
#   2. PMC passes (one counter per pass, kernel-trace only) of the network kernel at the bench batch
# Summaries are post-processed into profiles/ by scripts/summarize_profiles.py.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}   # default: the repo this script lives in
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" | cut -c1-300
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_net_once.py" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_net_once.py" > "$OUT/pmc_write.log" 2>&1
# the launch shape of the dual-stream search: 2048 evaluations, 4-wave workgroups
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_half" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_net_once.py" b6c64 2048 half > "$OUT/pmc_fetch_half.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_half" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_net_once.py" b6c64 2048 half > "$OUT/pmc_write_half.log" 2>&1
find "$OUT" -name "*.csv" | head -20
