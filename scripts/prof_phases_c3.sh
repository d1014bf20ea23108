#!/bin/bash
# kernel-only durations (rocprofv3) of the 10x128 network kernel at the C3 launch shape, truncated after each phase
# (LZ_NET_DEBUG_STOP: 1 staging, 2 stem, 3 trunk, 4 head convs, 5 policy head, 0 everything)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for stop in 1 2 3 4 5 0; do
  rm -rf gpurun_out/phc3_$stop
  LZ_NET_DEBUG_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/phc3_$stop -- python3 scripts/prof_net_once.py b10c128 16384 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/phc3_$stop/*/*kernel_stats.csv')[0])))
for r in rows:
    if 'net_forward' in r['Name']:
        print("stop=$stop avg_us=%.1f min_us=%.1f calls=%s" % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, r['Calls']))
PY
  rm -rf gpurun_out/phc3_$stop
done
