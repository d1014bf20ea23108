"""Same-box A/B of the tree step: the one-wave kernel (LZ_TREE_SPLIT=0) against the two-waves-per-game kernel (round 6) at
the launch shapes of the bench -- per-launch time of `tree_expand_select_kernel` (HIP events around every launch, direct
launches, halves serialised: what rocprofv3 reports) and the whole C2 / C3 step.  One child process per setting.
The switch under test is LZ_AB_VAR (default LZ_TREE_SPLIT; LZ_TREE_F32SEL: the fp32 argmax of the descent).
usage: [LZ_AB_VAR=LZ_TREE_F32SEL] python scripts/micro/tree_ab.py [C2] [C3]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VAR = os.environ.get("LZ_AB_VAR", "LZ_TREE_SPLIT")
for wl, steps in (("C2", 100), ("C3", 6)):
    if len(sys.argv) > 1 and wl not in sys.argv[1:]:
        continue
    for rep in range(2):
        for split in ("0", "1"):
            env = dict(os.environ)
            env[VAR] = split
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", str(steps), "--warmup", "3",
                                "--also", "none", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(json.dumps({"workload": wl, VAR: split, "failed": r.stderr[-400:]}), flush=True)
                continue
            d = json.loads(line[-1])
            sec = (d["roofline"].get("secondary") or {}).get("tree_expand_select_kernel", {})
            print(json.dumps({"workload": wl, VAR: split, "positions_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                              "tree_us_as_scheduled": (sec.get("as_scheduled") or {}).get("avg_launch_us"),
                              "tree_us_serialized": (sec.get("serialized") or {}).get("avg_launch_us"),
                              "tree_hbm_frac_serialized": (sec.get("serialized") or {}).get("frac"),
                              "sclk": (d.get("clocks") or {}).get("sclk_mhz_mean")}), flush=True)
