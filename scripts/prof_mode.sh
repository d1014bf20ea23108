#!/bin/bash
# rocprofv3 kernel stats of a short bench run in another mode, summarised into gpurun_out/<tag>_kernel_stats.md (scratch:
# copy the summary you want judged into profiles/)
#   bash scripts/prof_mode.sh r01_bench_root --search root --steps 30 --warmup 5
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}   # default: the repo this script lives in
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="$*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - <<PY
import csv, glob, json
rows = list(csv.DictReader(open(sorted(glob.glob("$OUT/*/*kernel_stats.csv"))[-1])))
line = [l for l in open("$OUT/bench.log") if l.startswith("{")][-1]
b = json.loads(line)
with open("gpurun_out/${TAG}_kernel_stats.md", "w") as f:
    f.write("# $TAG: rocprofv3 --kernel-trace --stats of \`python3 bench.py --no-cpu-baseline $ARGS\`\n\n")
    f.write(f"workload: {b['config']['workload']}\n\n")
    f.write(f"bench line of the same (profiled) run: value={b['value']} {b['unit']}, ms_per_step={b['ms_per_step']}, "
            f"roofline.frac={b['roofline']['frac']}, avg_launch_us={b['roofline'].get('avg_launch_us')}\n\n")
    f.write("The run includes the population's random pre-roll (120 plies of small ATen kernels), which is not part of the timed steps.\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:14]:
        f.write(f"| \`{r['Name'][:70]}\` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | {r['Percentage']} |\n")
print(open("gpurun_out/${TAG}_kernel_stats.md").read()[:1500])
PY
