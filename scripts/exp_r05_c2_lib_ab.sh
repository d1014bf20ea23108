cd "${GRAFT_REPO_ROOT}"
for rep in 1 2; do
for lib in variants/libheads_old.so liuzhou_amd/libliuzhou_hip.so; do
  LZ_HIP_LIB=$PWD/$lib python bench.py --workload C2 --steps 80 --warmup 5 --also none --no-cpu-baseline --no-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'positions/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done
