for c in 0 1; do
  LZ_TREE_COMPACT=$c python bench.py --workload C3 --steps 5 --warmup 2 --also none --no-cpu-baseline > gpurun_out/cmp_c3_$c.json 2>/dev/null
  LZ_TREE_COMPACT=$c python bench.py --workload C2 --steps 40 --warmup 5 --also none --no-cpu-baseline > gpurun_out/cmp_c2_$c.json 2>/dev/null
done
python - <<'PY'
import json
for w in ("c3","c2"):
    for c in (0,1):
        d=json.loads(open(f"gpurun_out/cmp_{w}_{c}.json").read().strip().splitlines()[-1])
        sec=(d["roofline"].get("secondary") or {}).get("tree_expand_select_kernel",{})
        print(w, "compact",c, "value",d["value"],"ms/step",d["ms_per_step"],"leaf_evals/s",d["leaf_evals_per_sec"], "tree us", {k:v.get("avg_launch_us") for k,v in sec.items() if isinstance(v,dict)}, "net probe us", d["roofline"].get("kernel_probe",{}).get("avg_launch_us"))
PY
