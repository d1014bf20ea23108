#!/bin/bash
# round 6, GPU call 9: the whole GPU suite + the timing guards on the final code, then the sustained windows
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_final2.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r06_gputest_final2.log; tail -4 gpurun_out/r06_gputest_final2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -m pytest tests -m perf -q > gpurun_out/r06_perf_guards.log 2>&1; tail -3 gpurun_out/r06_perf_guards.log
timeout -k 10 300 python bench.py --workload C2 --steps 2000 --warmup 5 --also none --no-cpu-baseline > gpurun_out/r06_bench_c2_sustained_2000_steps.json 2> gpurun_out/_s.err || { tail -5 gpurun_out/_s.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_c2_sustained_2000_steps.json').read().strip().splitlines()[-1]); print('C2 sustained', d['value'], d['ms_per_step'], d['roofline']['frac'], d['streams'], d['reuse']['pruned'], d['reuse']['dropped'])"
timeout -k 10 500 python bench.py --workload C3 --steps 100 --warmup 5 --also none --no-cpu-baseline > gpurun_out/r06_bench_c3_sustained_100_steps.json 2> gpurun_out/_s.err || { tail -5 gpurun_out/_s.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_c3_sustained_100_steps.json').read().strip().splitlines()[-1]); print('C3 sustained', d['value'], d['ms_per_step'], d['roofline']['frac'], d['clocks'].get('sclk_mhz_mean'), d['clocks'].get('power_w_mean'), d['reuse']['pruned'], d['reuse']['dropped'])"
