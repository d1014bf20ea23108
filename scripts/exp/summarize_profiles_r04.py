#!/usr/bin/env python3
"""Round-4 tracked profile summaries from gpurun_out/prof_r04/ (scripts/exp/prof_r04.sh):
  profiles/r04_bench_c3_kernel_stats.md, r04_bench_c2_kernel_stats.md   kernel-stats tables of bench runs taken under the
       bench protocol, each with the bench line, package power / sclk of THE SAME run, and the reconciliation
       (sims + 1) x AverageNs(network kernel) vs ms_per_step
  profiles/r04_pmc_net_forward.md + profiles/traffic.json               PMC traffic per launch shape
  profiles/r04_pmc_sq_net_forward.md                                    SQ counters of the two production shapes"""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"        # `summarize_profiles_r04.py r05`: the same tables for round 5
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")


def hits(pattern):
    return sorted(glob.glob(os.path.join(src, pattern)))


def bench_table(sub, title, launches_per_step):
    fs = hits(f"{sub}/*/*_kernel_stats.csv")
    log = os.path.join(src, f"{sub}.log")
    if not fs or not os.path.exists(log):
        print("missing", sub)
        return
    rows = list(csv.DictReader(open(fs[0])))
    bj = json.loads([l for l in open(log) if l.startswith("{")][-1])
    kp = bj["roofline"].get("kernel_probe", {})
    ck = bj.get("clocks") or {}
    net = [r for r in rows if "net_forward_kernel" in r["Name"]]
    tree = [r for r in rows if "tree_expand_select_kernel<false" in r["Name"] or "tree_expand_select_kernelILb0" in r["Name"]]
    with open(os.path.join(out, f"{tag}_{sub}_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `{title}`\n\n{bj['config']['workload']}\n\n")
        f.write(f"Bench line of the same (profiled) run: value = {bj['value']} {bj['unit']}, ms_per_step = {bj['ms_per_step']} "
                f"over {bj['steps']} timed steps after {bj['soak']['steps']} soak steps ({bj['soak']['seconds']} s) + {bj['warmup']} "
                f"warm-up steps, roofline.achieved = {bj['roofline']['achieved']} TFLOP/s = {bj['roofline']['frac']} (timed "
                f"schedule), kernel_probe.avg_launch_us = {kp.get('avg_launch_us')} (HIP events; serialized: "
                f"{kp.get('avg_launch_us_serialized')}), profiler_attached = {bj.get('profiler_attached')}.\n\n"
                f"Package during the timed steps of THIS run (in-process sysfs sampler, {ck.get('samples_in_timed_region')} samples): "
                f"power mean {ck.get('power_w_mean')} W (min {ck.get('power_w_min')}), sclk mean {ck.get('sclk_mhz_mean')} MHz "
                f"(min {ck.get('sclk_mhz_min')}, max {ck.get('sclk_mhz_max')}).\n\n")
        if net:
            avg_us = float(net[0]["AverageNs"]) / 1e3
            per_step_ms = launches_per_step * avg_us / 1e3
            f.write(f"**Reconciliation.**  {launches_per_step} network launches per step x AverageNs {avg_us:.2f} us = "
                    f"{per_step_ms:.2f} ms of network kernel per step"
                    + (f" (+ {launches_per_step - 1} x {float(tree[0]['AverageNs']) / 1e3:.2f} us of tree kernel = "
                       f"{(launches_per_step - 1) * float(tree[0]['AverageNs']) / 1e6:.2f} ms)" if tree else "")
                    + f" against this run's ms_per_step = {bj['ms_per_step']}"
                    + (" -- the two streams of C2 overlap the halves' kernels, the trace lists them one by one" if bj["config"].get("dual_stream") else "")
                    + f".  AverageNs covers every launch of the process (soak, warm-up, timed steps and the 3 probe steps).\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:14]:
            f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |\n")
        reuse = bj.get("reuse") or {}
        f.write(f"\nTree arenas of the run: {json.dumps(reuse.get('edge_pool'))}; pruned {reuse.get('pruned')}, dropped "
                f"{reuse.get('dropped')} in about {reuse.get('moves_searched_about')} searched moves.\n")
    print(sub, "value", bj["value"], "ms/step", bj["ms_per_step"], "net avg us", float(net[0]["AverageNs"]) / 1e3 if net else None)


def pmc_mean(sub, counter):
    vals, res = [], set()
    fs = hits(f"{sub}/*/*_counter_collection.csv")
    if not fs:
        return None, 0, []
    for r in csv.DictReader(open(fs[0])):
        if "net_forward_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            res.add((r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"]))
    return (sum(vals) / max(1, len(vals)) if vals else None), len(vals), sorted(res)


def traffic():
    tj = {"_note": "HBM-side bytes of one net_forward_kernel launch (rocprofv3 --pmc, separate passes): 2*FETCH_SIZE (gfx950 "
                   f"half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB->bytes. Source: profiles/{tag}_pmc_net_forward.md"}
    lines = []
    WBYTES = {"b6c64": 0.99e6, "b10c128": 5.9e6}
    for model, n, shape in (("b10c128", 16384, "full"), ("b6c64", 4096, "full"), ("b6c64", 2048, "half")):
        name = f"{model}_B{n}_{shape}"
        fetch, _, res = pmc_mean(f"pmc_fetch_{name}", "FETCH_SIZE")
        write, _, _ = pmc_mean(f"pmc_write_{name}", "WRITE_SIZE")
        if fetch is None or write is None:
            print("missing pmc", name)
            continue
        t = int(2 * fetch * 1024 + write * 1024)
        key = f"net_forward_{model}_B{n}" + ("_half" if shape == "half" else "")
        tj[key] = t
        algo = n * (32 + 436) + 8 * WBYTES[model]
        scratch, vgpr, agpr, lds = res[0]
        lines.append(f"| `{key}` | {fetch:.1f} | {write:.1f} | {t / 1e6:.2f} MB | {algo / 1e6:.2f} MB | {t / algo:.2f}x | "
                     f"{scratch} B/lane, {vgpr} VGPR + {agpr} AGPR, LDS {lds} B |")
    if len(tj) < 4:
        return
    with open(os.path.join(out, f"{tag}_pmc_net_forward.md"), "w") as f:
        f.write(f"# {tag} PMC passes of `net_forward_kernel` at the three launch shapes of bench.py\n\n"
                f"One counter per pass with `--kernel-trace` only (HBM section of MI355X_MICROARCH.md), `scripts/exp/prof_{tag}.sh`:\n\n"
                "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_net_once.py <model> <batch> <full|half>\n"
                "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_net_once.py <model> <batch> <full|half>\n\n"
                "Launch = one network evaluation of the whole batch from 32-byte packed states (the search loop's shape), mean "
                "over 20 launches.  traffic = 2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE.  Algorithmic bytes = "
                "batch x (32 B state in + 436 B heads / value out) + the packed weights once per XCD L2 (8 x).\n\n"
                "| bench.py key | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic / launch | algorithmic | ratio | kernel resources |\n"
                "|---|---:|---:|---:|---:|---:|---|\n" + "\n".join(lines) + "\n")
    json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(tj))


def sq():
    cols = {}
    for name, label, evals, flops in (("b10c128_B16384_full", "`<128,8,8>` x 16 384 (C3 launch)", 16384, 214.54e6),
                                      ("b6c64_B2048_half", "`<64,8,4>` x 2 048 (C2 half launch)", 2048, 33.01e6)):
        acc = collections.defaultdict(list)
        for sub in (f"sq_{name}", f"sq2_{name}"):
            for f in hits(f"{sub}/*/*_counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    if "net_forward_kernel" in r["Kernel_Name"]:
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if acc:
            cols[label] = ({k: sum(v) / len(v) for k, v in acc.items()}, evals, flops)
    if not cols:
        return
    keys = sorted(set(k for c, _, _ in cols.values() for k in c))
    with open(os.path.join(out, f"{tag}_pmc_sq_net_forward.md"), "w") as g:
        g.write(f"# {tag}: SQ counters of the network kernel at the two production launch shapes\n\n"
                f"`scripts/exp/prof_{tag}.sh`: two `rocprofv3 --pmc ... --kernel-trace` passes (no other trace domain) per shape of "
                "`scripts/prof_net_once.py` (20 launches alone on the device; means per launch).\n\n"
                "| counter | " + " | ".join(cols) + " |\n|---|" + "---:|" * len(cols) + "\n")
        for k in keys:
            g.write(f"| {k} | " + " | ".join(f"{c.get(k, float('nan')):.5g}" for c, _, _ in cols.values()) + " |\n")
        g.write("\nDerived:\n\n")
        for label, (c, evals, flops) in cols.items():
            if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
                continue
            simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            dense = flops / (2 * 16 * 16 * 32)
            g.write(f"* {label}: launch = GRBM_GUI_ACTIVE / 8 = {c['GRBM_GUI_ACTIVE'] / 8.0:.4g} shader cycles; matrix pipes busy "
                    f"**{c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f}** of the SIMD-cycles; MFMAs per evaluation "
                    f"{c.get('SQ_INSTS_MFMA', 0) / evals:.0f} of {dense:.0f} dense; LDS instructions per evaluation "
                    f"{c.get('SQ_INSTS_LDS', 0) / evals:.0f}, bank-conflict cycles per LDS instruction "
                    f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_INSTS_LDS', 1), 1):.2f}; waves: "
                    f"{c.get('SQ_ACTIVE_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f} issuing, "
                    f"{c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f} issue-stalled, "
                    f"{c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f} waiting.\n")
    print("sq ok")


if __name__ == "__main__":
    os.makedirs(out, exist_ok=True)
    bench_table("bench_c3", "python3 bench.py --steps 10 --warmup 5 --also none --no-cpu-baseline", 801)
    bench_table("bench_c2", "python3 bench.py --workload C2 --steps 100 --warmup 5 --also none --no-cpu-baseline", 402)
    traffic()
    sq()
