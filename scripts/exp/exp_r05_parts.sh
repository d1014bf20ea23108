#!/bin/bash
# Round 5: C2 with the games split over 2 / 3 / 4 streams, with the runtime's default 4 hardware queues and with 8.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {
  python bench.py --workload C2 --steps 60 --warmup 5 --also none --no-cpu-baseline --no-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts', '$1', 'queues', '$2', 'positions/s', d['value'], 'ms/step', d['ms_per_step'])"
}
for q in 4 8; do
  for parts in 2 3 4; do
    GPU_MAX_HW_QUEUES=$q LZ_TREE_PARTS=$parts run $parts $q
  done
done
