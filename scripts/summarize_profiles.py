#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/prof_round.sh (gpurun_out/prof_<tag>/) into the tracked summaries
under profiles/: kernel-stats table (csv + md), the PMC traffic note and profiles/traffic.json (read by bench.py)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)))
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


rows = list(csv.DictReader(open(one("bench/*/*_kernel_stats.csv"))))
with open(os.path.join(out, f"{name}_bench_tree_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:40]:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                    r["MaxNs"], r["StdDev"]])
bench_line = [l for l in open(os.path.join(src, "bench.log")) if l.startswith("{")][-1]
bj = json.loads(bench_line)
with open(os.path.join(out, f"{name}_bench_tree_kernel_stats.md"), "w") as f:
    f.write(f"# {name}: rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline` "
            f"({bj['config']['workload']}; timed steps replay a hipGraph, the roofline probe launches directly)\n\n")
    f.write(f"bench line of the same run: value={bj['value']} {bj['unit']}, ms_per_step={bj['ms_per_step']}, "
            f"roofline.avg_launch_us={bj['roofline']['avg_launch_us']}, "
            f"avg_launch_us_serialized={bj['roofline'].get('avg_launch_us_serialized')} (HIP events) vs the AverageNs below\n\n")
    if bj["roofline"].get("streams", 1) == 2:
        f.write("The default search runs two half-batches on two streams so that the tree kernel of one overlaps the network "
                "kernel of the other.  rocprofv3's kernel trace serialises the streams: this profiled run is slower than an "
                "unprofiled one (no overlap), and the durations below are those of launches running alone -- compare them "
                "with `avg_launch_us_serialized`; in an unprofiled run `avg_launch_us` is longer because each launch shares "
                "the chip with the other half's kernels.\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:16]:
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                f"{float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |\n")


def pmc_mean(sub, counter):
    vals, scratch = [], set()
    for r in csv.DictReader(open(one(f"{sub}/*/*_counter_collection.csv"))):
        if "net_forward_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            scratch.add((r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"]))
    return sum(vals) / max(1, len(vals)), len(vals), scratch


fetch, n1, regs = pmc_mean("pmc_fetch", "FETCH_SIZE")
write, n2, _ = pmc_mean("pmc_write", "WRITE_SIZE")
traffic = int(2 * fetch * 1024 + write * 1024)
try:
    fetch_h, _, regs_h = pmc_mean("pmc_fetch_half", "FETCH_SIZE")
    write_h, _, _ = pmc_mean("pmc_write_half", "WRITE_SIZE")
    traffic_h = int(2 * fetch_h * 1024 + write_h * 1024)
except SystemExit:
    fetch_h = write_h = 0.0
    traffic_h = None
scratch, vgpr, agpr, lds = sorted(regs)[0]
with open(os.path.join(out, f"{name}_pmc_net_forward.md"), "w") as f:
    f.write(f"# {name} PMC passes: net_forward_kernel<64,16>, 4096 evaluations per launch (the bench batch)\n\n"
            "One counter per pass with `--kernel-trace` only (HBM section of MI355X_MICROARCH.md), "
            "`scripts/prof_round.sh`:\n\n"
            "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_net_once.py\n"
            "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_net_once.py\n\n"
            f"| counter | mean over {n1} launches | corrected bytes / launch |\n|---|---:|---:|\n"
            f"| FETCH_SIZE | {fetch:.1f} KiB | x2 (gfx950 half-count correction) = {2 * fetch * 1024 / 1e6:.1f} MB |\n"
            f"| WRITE_SIZE | {write:.1f} KiB | exact = {write * 1024 / 1e6:.1f} MB |\n"
            f"| traffic (bench.py `roofline.traffic`) | | {traffic / 1e6:.1f} MB |\n"
            + (f"| dual-stream launch shape (<64,8,4>, 2048 evaluations): FETCH_SIZE {fetch_h:.1f} KiB, WRITE_SIZE {write_h:.1f} KiB "
               f"| | {traffic_h / 1e6:.1f} MB |\n" if traffic_h else "") + "\n"
            f"Kernel resources reported by the trace: scratch {scratch} B/lane, {vgpr} VGPR + {agpr} AGPR, LDS {lds} B/WG.\n\n"
            "Algorithmic HBM bytes of the launch: 4096 x 1584 B float planes in (32 B packed states on the tree path) "
            "+ 4096 x 436 B out = 8.3 MB, plus the packed weights once per XCD L2 (8 x 0.99 MB).  Traffic above that is "
            "register-spill scratch; the kernel is MFMA-bound, so it costs latency in the prologue / head phases, not "
            "bandwidth.\n")
tj = {"_note": "HBM-side bytes of one net_forward_kernel launch (rocprofv3 --pmc, separate passes): 2*FETCH_SIZE "
               "(gfx950 half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB->bytes. Source: profiles/"
               f"{name}_pmc_net_forward.md",
      "net_forward_b6c64_B4096": traffic,
      "net_forward_b6c64_B2048_half": traffic_h,
      "raw": {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches": n1}}
json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(f"net_forward avg {float(rows[0]['AverageNs']) / 1e3:.1f} us; traffic {traffic / 1e6:.1f} MB; scratch {scratch} B/lane")
