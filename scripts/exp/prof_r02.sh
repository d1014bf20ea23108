#!/bin/bash
# Profiles behind bench.py's roofline numbers, round 2 (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the headline workload C3 and of C2 (bench.py itself, short)
#   2. PMC passes (one counter per pass, --kernel-trace only) of the network kernel at the three launch shapes of
#      the bench: <128,8,8> x 16384 evaluations (C3), <64,16,8> x 4096 and <64,8,4> x 2048 (C2, one / two streams)
# Summaries are post-processed into profiles/ by scripts/exp/summarize_profiles_r02.py.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c3" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --soak-seconds 0 --also none --no-cpu-baseline > "$OUT/bench_c3.log" 2>&1
tail -1 "$OUT/bench_c3.log" | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c2" -- python3 "$ROOT/bench.py" --workload C2 --steps 8 --warmup 2 --soak-seconds 0 --no-cpu-baseline > "$OUT/bench_c2.log" 2>&1
tail -1 "$OUT/bench_c2.log" | cut -c1-200
for shape in "b10c128 16384 full" "b6c64 4096 full" "b6c64 2048 half"; do
  set -- $shape
  name="$1_B$2_$3"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$name" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/pmc_fetch_$name.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$name" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/pmc_write_$name.log" 2>&1
  echo "pmc $name done"
done
find "$OUT" -name "*.csv" | wc -l; du -sh "$OUT"
