#!/bin/bash
# Round 5: SQ counters of the tree kernels at the C2 half-launch shape (2 048 games, 6x64, 200 sims) and the C3 shape
# (16 384 games, 10x128, 800 sims).  Separate --pmc passes with --kernel-trace only; the program itself after `--`.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r05_tree
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for shape in "b6c64 2048 200 3" "b10c128 16384 800 2"; do
  set -- $shape
  tag="$1_$2"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$tag" -- python3 "$ROOT/scripts/prof_tree_once.py" $1 $2 $3 $4 > "$OUT/stats_$tag.log" 2>&1
  echo "stats $tag done"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$OUT/pmc1_$tag" -- python3 "$ROOT/scripts/prof_tree_once.py" $1 $2 $3 $4 > "$OUT/pmc1_$tag.log" 2>&1
  echo "pmc1 $tag done"
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU \
      --kernel-trace --output-format csv -d "$OUT/pmc2_$tag" -- python3 "$ROOT/scripts/prof_tree_once.py" $1 $2 $3 $4 > "$OUT/pmc2_$tag.log" 2>&1
  echo "pmc2 $tag done"
  rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_SENDMSG SQ_INSTS_BRANCH \
      --kernel-trace --output-format csv -d "$OUT/pmc3_$tag" -- python3 "$ROOT/scripts/prof_tree_once.py" $1 $2 $3 $4 > "$OUT/pmc3_$tag.log" 2>&1
  echo "pmc3 $tag done (optional counters; a refused name leaves this pass empty)"
done
python3 "$ROOT/scripts/exp/summarize_profiles_r05.py" "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
# the raw traces are tens of MB: keep the summary, the logs and the kernel-stats tables only
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
