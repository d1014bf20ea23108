#!/bin/bash
# Round 5: same-box A/B of the C2 bench between builds of the HIP library (LZ_HIP_LIB), two rounds each.
# usage: bash scripts/exp/exp_r05_c2_lib_ab.sh path/to/old.so [path/to/new.so ...]   (paths relative to the repository root;
# the in-tree library is liuzhou_amd/libliuzhou_hip.so; an older build: `git stash` / checkout, python -m liuzhou_amd.build, cp)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=(liuzhou_amd/libliuzhou_hip.so)
for rep in 1 2; do
  for lib in "${LIBS[@]}"; do
    LZ_HIP_LIB=$PWD/$lib python bench.py --workload C2 --steps 80 --warmup 5 --also none --no-cpu-baseline --no-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'positions/s', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
