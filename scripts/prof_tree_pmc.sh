#!/bin/bash
# instruction-mix counters of the fused tree kernel (one PMC pass, kernel-trace only)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}   # default: the repo this script lives in
OUT=$PWD/gpurun_out/prof_tree_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/bench.log" 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tree_expand_select_kernel<false>" in k or "net_forward" in k:
            acc[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-18s mean %.1f over %d launches" % (c, sum(v) / len(v), len(v)))
PY
