#!/bin/bash
# SQ counters of the network kernel at the C3 launch shape (one PMC pass, --kernel-trace only): MFMA pipe busy cycles,
# wave cycles, wait buckets, LDS instructions / bank conflicts -- the MFMA-utilisation evidence for DESIGN.md section 5.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_sq_net
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > "$OUT/mfma_counters.txt"
for shape in "b10c128 16384 full" "b6c64 4096 full"; do
  set -- $shape
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
      --kernel-trace --output-format csv -d "$OUT/sq_$1" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/sq_$1.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$OUT/sq2_$1" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/sq2_$1.log" 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/sq*_*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "net_forward_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.rstrip("/").split("/")[-1])
    for c, v in sorted(acc.items()):
        print("   %-32s mean %.4g over %d launches" % (c, sum(v) / len(v), len(v)))
PY
cat "$OUT/mfma_counters.txt"
