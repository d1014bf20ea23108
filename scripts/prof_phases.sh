#!/bin/bash
# kernel-only durations (rocprofv3) of the network kernel truncated after each phase (LZ_NET_DEBUG_STOP)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}   # default: the repo this script lives in
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for stop in 1 2 3 4 5 0; do
  rm -rf gpurun_out/ph_$stop
  LZ_NET_DEBUG_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph_$stop -- python3 scripts/prof_net_once.py b6c64 4096 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/ph_$stop/*/*kernel_stats.csv')[0])))
for r in rows:
    if 'net_forward' in r['Name']:
        print("stop=$stop avg_us=%.1f min_us=%.1f calls=%s" % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, r['Calls']))
PY
done
