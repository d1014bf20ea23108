# C3 harness with the launch form forced (LZ_TREE_COMPACT=1: lists in every search) and automatic (dense in steady state)
for c in auto 1; do
  if [ "$c" = "auto" ]; then unset LZ_TREE_COMPACT; else export LZ_TREE_COMPACT=$c; fi
  python bench.py --workload C3 --steps 5 --warmup 2 --also none --no-cpu-baseline > gpurun_out/cmp_c3_$c.json 2>/dev/null
done
unset LZ_TREE_COMPACT
python - <<'PY'
import json
for c in ("auto","1"):
    d=json.loads(open(f"gpurun_out/cmp_c3_{c}.json").read().strip().splitlines()[-1])
    sec=(d["roofline"].get("secondary") or {}).get("tree_expand_select_kernel",{})
    print("c3 compact",c, "value",d["value"],"ms/step",d["ms_per_step"],"leaf_evals/s",d["leaf_evals_per_sec"], "tree us", {k:v.get("avg_launch_us") for k,v in sec.items() if isinstance(v,dict)}, "net probe us", d["roofline"].get("kernel_probe",{}).get("avg_launch_us"))
PY
