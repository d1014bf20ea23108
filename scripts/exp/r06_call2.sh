#!/bin/bash
# round 6, GPU call 2: stream-pair A/B inside bench.py (C2), host-I/O rehearsal on the GPU box, full default bench
set -o pipefail
mkdir -p gpurun_out
for m in probe priority probe priority; do
  LZ_STREAM_PAIR=$m timeout -k 10 150 python bench.py --workload C2 --steps 150 --warmup 5 --also none --no-cpu-baseline \
      > gpurun_out/_b.json 2> gpurun_out/_b.err || { tail -5 gpurun_out/_b.err; exit 1; }
  python - <<'PY' >> gpurun_out/r06_stream_pair_bench.jsonl
import json
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(json.dumps({"pair": d["streams"], "value": d["value"], "ms": d["ms_per_step"], "clocks": d.get("clocks")}))
PY
done
cat gpurun_out/r06_stream_pair_bench.jsonl
for args in "--crc32 1" "--crc32 0" "--crc32 1 --writers 3" "--crc32 0 --writers 3"; do
  timeout -k 10 200 python scripts/rehearse_host_io.py --procs 8 --gb-per-proc 3 --rate-gbps 1.4 $args >> gpurun_out/r06_host_io_gpubox.jsonl || exit 1
done
cut -c1-600 gpurun_out/r06_host_io_gpubox.jsonl
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default_1.json 2> gpurun_out/r06_bench_default_1.err
echo "bench rc=$?"
tail -c 1800 gpurun_out/r06_bench_default_1.json
