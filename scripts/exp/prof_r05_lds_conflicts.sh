#!/bin/bash
# Round 5: where the LDS bank conflicts of net_forward_kernel<64,8,4> (C2 half launch, 2 048 positions) come from: the kernel
# truncated after each phase (LZ_NET_DEBUG_STOP: 1 staging, 2 stem, 3 trunk, 4 head convs, 5 policy head, 0 everything) under
# one --pmc pass (kernel-trace only); differences between consecutive rows = the phase's own counts.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for stop in 1 2 3 4 5 0; do
  rm -rf gpurun_out/ldsc_$stop
  LZ_NET_DEBUG_STOP=$stop rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/ldsc_$stop -- python3 scripts/prof_net_once.py b6c64 2048 half > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/ldsc_$stop/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'net_forward' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print("stop=$stop " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k,v in sorted(acc.items())))
PY
  rm -rf gpurun_out/ldsc_$stop
done
