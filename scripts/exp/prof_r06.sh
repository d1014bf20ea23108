#!/bin/bash
# Round-6 profiles (run on the GPU box from the repo root): the kernel traces are taken UNDER THE BENCH PROTOCOL -- soak on,
# >= 10 timed steps, the in-process clock sampler running (bench.py keeps it on under rocprofv3 since round 4) -- so that
# (sims + 1) x AverageNs of the network kernel can be set against the SAME run's ms_per_step, power and sclk.
#   1. rocprofv3 --kernel-trace --stats of C3 (10 timed steps) and C2 (100 timed steps)
#   2. PMC traffic of the network kernel at the three launch shapes (one counter per pass, --kernel-trace only)
#   3. SQ counters of the two production shapes (<128,8,8> x 16384, <64,8,4> x 2048)
# Summaries: scripts/exp/summarize_profiles_r04.py r05 -> profiles/r06_*.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r06
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c3" -- python3 "$ROOT/bench.py" --steps 10 --warmup 5 --also none --no-cpu-baseline > "$OUT/bench_c3.log" 2>&1
tail -1 "$OUT/bench_c3.log" | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c2" -- python3 "$ROOT/bench.py" --workload C2 --steps 100 --warmup 5 --also none --no-cpu-baseline > "$OUT/bench_c2.log" 2>&1
tail -1 "$OUT/bench_c2.log" | cut -c1-200
# big traces are summarised by rocprofv3 itself (*_kernel_stats.csv); the per-dispatch trace is not needed afterwards
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
for shape in "b10c128 16384 full" "b6c64 4096 full" "b6c64 2048 half"; do
  set -- $shape
  name="$1_B$2_$3"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$name" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/pmc_fetch_$name.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$name" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/pmc_write_$name.log" 2>&1
  echo "pmc $name done"
done
for shape in "b10c128 16384 full" "b6c64 2048 half"; do
  set -- $shape
  name="$1_B$2_$3"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
      --kernel-trace --output-format csv -d "$OUT/sq_$name" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/sq_$name.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$OUT/sq2_$name" -- python3 "$ROOT/scripts/prof_net_once.py" $1 $2 $3 > "$OUT/sq2_$name.log" 2>&1
  echo "sq $name done"
done
python3 "$ROOT/scripts/exp/summarize_profiles_r04.py" r06 > "$OUT/summary.log" 2>&1 || tail -5 "$OUT/summary.log"
cat "$OUT/summary.log" | tail -12
mkdir -p "$OUT/summary" && cp "$ROOT"/profiles/r06_* "$ROOT/profiles/traffic.json" "$OUT/summary/" 2>/dev/null
find "$OUT" -type f -size +2M -delete
find "$OUT" -name "*.csv" | wc -l; du -sh "$OUT"
