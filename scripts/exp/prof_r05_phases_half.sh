#!/bin/bash
# Round 5: kernel-only durations (rocprofv3) of net_forward_kernel<64,8,4> at the C2 half-launch shape (2 048 positions)
# truncated after each phase (LZ_NET_DEBUG_STOP: 1 staging, 2 stem, 3 trunk, 4 head convs, 5 policy head, 0 everything)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for stop in 1 2 3 4 5 0; do
  rm -rf gpurun_out/phh_$stop
  LZ_NET_DEBUG_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/phh_$stop -- python3 scripts/prof_net_once.py b6c64 2048 half > /dev/null 2>&1
  python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/phh_$stop/*/*kernel_stats.csv')[0])))
for r in rows:
    if 'net_forward' in r['Name']:
        print("stop=$stop avg_us=%.1f min_us=%.1f calls=%s" % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, r['Calls']))
PY
  rm -rf gpurun_out/phh_$stop
done
