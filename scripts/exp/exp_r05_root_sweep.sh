python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "root_puct" > gpurun_out/t11.log 2>&1; tail -4 gpurun_out/t11.log
for sims in 1024 8192 65536; do
  for bin in 1 0; do
    LZ_ROOT_PUCT_BIN=$bin python bench.py --search root --games 16384 --sims $sims --model b10c128 --steps 10 --warmup 3 --also none --no-cpu-baseline --no-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sims',$sims,'binned',$bin,'positions/s',d['value'],'ms/step',d['ms_per_step'])"
  done
done
