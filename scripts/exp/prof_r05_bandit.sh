#!/bin/bash
# Round 5: SQ counters of the root bandit kernel (root_puct_binned_kernel) at 16 384 roots x 8 192 pulls, 10x128 net:
# two --pmc passes with --kernel-trace only, the program itself after `--`.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r05_bandit
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--search root --games 16384 --sims 8192 --model b10c128 --steps 2 --warmup 1 --soak-seconds 0 --also none --no-cpu-baseline --no-probe --no-smi"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/pmc1" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU \
    --kernel-trace --output-format csv -d "$OUT/pmc2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc2.log" 2>&1
python3 - <<PY
import collections, csv, glob
out = "$OUT"
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    print("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|")
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "root_puct_binned_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("\nroot_puct_binned_kernel, mean per launch over", len(next(iter(acc.values()), [])), "launches")
for k in sorted(m): print(f"- {k}: {m[k]:.4g}")
if "SQ_WAVES" in m:
    print(f"- waves that do pulls are a fraction of SQ_WAVES (the others find no job and leave); SQ_INSTS_VALU per wave {m['SQ_INSTS_VALU']/m['SQ_WAVES']:.0f}")
if "SQ_WAVE_CYCLES" in m:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
        if k in m: print(f"- {k} / SQ_WAVE_CYCLES: {m[k]/m['SQ_WAVE_CYCLES']:.3f}")
PY
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
