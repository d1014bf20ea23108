#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/exp/prof_r02.sh (gpurun_out/prof_<tag>/) into the tracked summaries under
profiles/: kernel-stats tables of the C3 and C2 bench runs, the PMC traffic note, and profiles/traffic.json (read by
bench.py for `roofline.traffic`)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)))
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


def bench_table(sub, title):
    rows = list(csv.DictReader(open(one(f"{sub}/*/*_kernel_stats.csv"))))
    bj = json.loads([l for l in open(os.path.join(src, f"{sub}.log")) if l.startswith("{")][-1])
    with open(os.path.join(out, f"{tag}_{sub}_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `{title}`\n\n{bj['config']['workload']}\n\n")
        kp = bj["roofline"].get("kernel_probe", {})
        f.write(f"bench line of the same (profiled) run: value = {bj['value']} {bj['unit']}, ms_per_step = {bj['ms_per_step']}, "
                f"roofline.achieved = {bj['roofline']['achieved']} TFLOP/s (timed schedule), kernel_probe.avg_launch_us = "
                f"{kp.get('avg_launch_us')} (HIP events; serialized: {kp.get('avg_launch_us_serialized')}) -- compare with the "
                "AverageNs of `net_forward_kernel` below.  rocprofv3's kernel trace serialises streams, so a two-stream "
                "run is slower under the profiler than unprofiled and its durations are those of launches running alone.\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:14]:
            f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |\n")
    net = [r for r in rows if "net_forward_kernel" in r["Name"]]
    return bj, (float(net[0]["AverageNs"]) / 1e3 if net else None)


def pmc_mean(sub, counter):
    vals, res = [], set()
    for r in csv.DictReader(open(one(f"{sub}/*/*_counter_collection.csv"))):
        if "net_forward_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            res.add((r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"]))
    return sum(vals) / max(1, len(vals)), len(vals), sorted(res)


tj = {"_note": "HBM-side bytes of one net_forward_kernel launch (rocprofv3 --pmc, separate passes): 2*FETCH_SIZE (gfx950 "
               f"half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB->bytes. Source: profiles/{tag}_pmc_net_forward.md"}
lines = []
FLOPS = {"b6c64": 33.01e6, "b10c128": 214.54e6}
WBYTES = {"b6c64": 0.99e6, "b10c128": 5.9e6}
for model, n, shape in (("b10c128", 16384, "full"), ("b6c64", 4096, "full"), ("b6c64", 2048, "half")):
    name = f"{model}_B{n}_{shape}"
    fetch, n1, res = pmc_mean(f"pmc_fetch_{name}", "FETCH_SIZE")
    write, _, _ = pmc_mean(f"pmc_write_{name}", "WRITE_SIZE")
    traffic = int(2 * fetch * 1024 + write * 1024)
    key = f"net_forward_{model}_B{n}" + ("_half" if shape == "half" else "")
    tj[key] = traffic
    algo = n * (32 + 436) + 8 * WBYTES[model]
    scratch, vgpr, agpr, lds = res[0]
    lines.append(f"| `{key}` | {fetch:.1f} | {write:.1f} | {traffic / 1e6:.2f} MB | {algo / 1e6:.2f} MB | "
                 f"{traffic / algo:.2f}x | {scratch} B/lane, {vgpr} VGPR + {agpr} AGPR, LDS {lds} B |")
with open(os.path.join(out, f"{tag}_pmc_net_forward.md"), "w") as f:
    f.write(f"# {tag} PMC passes of `net_forward_kernel` at the three launch shapes of bench.py\n\n"
            "One counter per pass with `--kernel-trace` only (HBM section of MI355X_MICROARCH.md), `scripts/exp/prof_r02.sh`:\n\n"
            "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_net_once.py <model> <batch> <full|half>\n"
            "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_net_once.py <model> <batch> <full|half>\n\n"
            "Launch = one network evaluation of the whole batch from 32-byte packed states (the search loop's shape), mean "
            "over 20 launches.  traffic = 2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE.  Algorithmic bytes = "
            "batch x (32 B state in + 436 B heads / value out) + the packed weights once per XCD L2 (8 x).\n\n"
            "| bench.py key | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic / launch | algorithmic | ratio | kernel resources |\n"
            "|---|---:|---:|---:|---:|---:|---|\n" + "\n".join(lines) + "\n")
json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for sub, title in (("bench_c3", "python3 bench.py --steps 3 --warmup 1 --soak-seconds 0 --also none --no-cpu-baseline"),
                   ("bench_c2", "python3 bench.py --workload C2 --steps 8 --warmup 2 --soak-seconds 0 --no-cpu-baseline")):
    bj, avg = bench_table(sub, title)
    print(sub, "value", bj["value"], "net avg us", avg)
print(json.dumps(tj, indent=1))
