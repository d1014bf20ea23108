#!/bin/bash
# Round-3 profiles (run on the GPU box from the repo root):
#   1. everything scripts/exp/prof_r02.sh collects, under tag r03 (kernel stats of the C3 / C2 bench runs, PMC traffic of the
#      network kernel at the three launch shapes) -> scripts/exp/summarize_profiles_r02.py r03
#   2. the persistent search kernel (opt-in, csrc/lz_search.hip): kernel stats of the C2 bench with LZ_TREE_PERSISTENT=1,
#      and SQ counters (MFMA pipe busy, wave / wait cycles, instruction mix) of tree_search_persistent_kernel in their
#      own --pmc passes with --kernel-trace only
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash "$ROOT/scripts/exp/prof_r02.sh" r03
OUT=$ROOT/gpurun_out/prof_r03
cd /tmp && export TMPDIR=/tmp
LZ_TREE_PERSISTENT=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c2_persistent" -- python3 "$ROOT/bench.py" --workload C2 --steps 8 --warmup 2 --soak-seconds 0 --also none --no-cpu-baseline > "$OUT/bench_c2_persistent.log" 2>&1
tail -1 "$OUT/bench_c2_persistent.log" | cut -c1-200
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$OUT/sq_persistent" -- python3 "$ROOT/scripts/prof_persistent_once.py" > "$OUT/sq_persistent.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/sq2_persistent" -- python3 "$ROOT/scripts/prof_persistent_once.py" > "$OUT/sq2_persistent.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$OUT/sq_c2half" -- python3 "$ROOT/scripts/prof_net_once.py" b6c64 2048 half > "$OUT/sq_c2half.log" 2>&1
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/sq*_*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "tree_search_persistent_kernel" in r["Kernel_Name"] or "net_forward_kernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(d.rstrip("/").split("/")[-1])
    for c, v in sorted(acc.items()):
        print("   %-60s %-32s mean %.5g over %d launches" % (c[0], c[1], sum(v) / len(v), len(v)))
PY
# summaries are made HERE (gpurun copies back at most 64 MiB of gpurun_out/): tracked files land in profiles/ of this
# copy of the repo and are mirrored under $OUT/summary; raw traces above 2 MiB are dropped afterwards
python3 "$ROOT/scripts/exp/summarize_profiles_r02.py" r03 > "$OUT/summary.log" 2>&1 || tail -5 "$OUT/summary.log"
python3 "$ROOT/scripts/exp/summarize_profiles_r03.py" > "$OUT/summary_r03.log" 2>&1 || tail -5 "$OUT/summary_r03.log"
mkdir -p "$OUT/summary" && cp "$ROOT"/profiles/r03_* "$ROOT/profiles/traffic.json" "$OUT/summary/" 2>/dev/null
find "$OUT" -type f -size +2M -delete
find "$OUT" -name "*.csv" | wc -l; du -sh "$OUT"
