#!/usr/bin/env python3
"""Round-3 additions to the tracked profile summaries (raw input: gpurun_out/prof_r03/, scripts/exp/prof_r03.sh):
  profiles/r03_bench_c2_persistent_kernel_stats.md   rocprofv3 --kernel-trace --stats of the C2 bench with the opt-in
                                                      persistent search kernel (LZ_TREE_PERSISTENT=1)
  profiles/r03_pmc_sq_persistent.md                   SQ counters of tree_search_persistent_kernel next to the stand-alone
                                                      network kernel at the C2 half-batch launch shape"""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "gpurun_out", "prof_r03")
out = os.path.join(ROOT, "profiles")


def first(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)))
    return hits[0] if hits else None


def kernel_stats():
    f = first("bench_c2_persistent/*/*_kernel_stats.csv")
    log = os.path.join(src, "bench_c2_persistent.log")
    if not f or not os.path.exists(log):
        return
    rows = list(csv.DictReader(open(f)))
    bj = json.loads([l for l in open(log) if l.startswith("{")][-1])
    kp = bj["roofline"].get("kernel_probe", {})
    with open(os.path.join(out, "r03_bench_c2_persistent_kernel_stats.md"), "w") as g:
        g.write("# r03: rocprofv3 --kernel-trace --stats of `LZ_TREE_PERSISTENT=1 python3 bench.py --workload C2 --steps 8 "
                "--warmup 2 --soak-seconds 0 --also none --no-cpu-baseline`\n\n" + bj["config"]["workload"] + "\n\n")
        g.write(f"bench line of the same (profiled) run: value = {bj['value']} {bj['unit']}, ms_per_step = {bj['ms_per_step']}, "
                f"roofline.achieved = {bj['roofline']['achieved']} TFLOP/s (timed schedule), kernel_probe.avg_launch_us = "
                f"{kp.get('avg_launch_us')} for {kp.get('evals_per_launch')} evaluations per launch (HIP events around the "
                "search kernel) -- compare with the AverageNs of `tree_search_persistent_kernel` below: one launch is the whole "
                "search of a move, 4 096 games x 201 network evaluations + 201 tree steps.\n\n")
        g.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:12]:
            g.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |\n")


def counters(sub, needle):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, max([len(v) for v in acc.values()] or [0])


def sq_table():
    p1, n1 = counters("sq_persistent", "tree_search_persistent_kernel")
    p2, _ = counters("sq2_persistent", "tree_search_persistent_kernel")
    h, n2 = counters("sq_c2half", "net_forward_kernel")
    if not p1:
        return
    p = dict(p1, **p2)
    with open(os.path.join(out, "r03_pmc_sq_persistent.md"), "w") as g:
        g.write("# r03: SQ counters of the persistent search kernel (`tree_search_persistent_kernel<64,8,4>`, opt-in)\n\n"
                "`scripts/exp/prof_r03.sh`: separate `rocprofv3 --pmc ... --kernel-trace` passes (no other trace domain) of "
                "`scripts/prof_persistent_once.py` (C2: 4 096 games, 200 simulations, direct launches; mean over "
                f"{n1} launches = searched moves) and, beside it, of the stand-alone network kernel at the C2 half-batch shape "
                f"(`prof_net_once.py b6c64 2048 half`, {n2} launches).\n\n"
                "| counter | persistent search kernel (per launch = 4 096 x 201 evaluations + tree steps) | "
                "`net_forward_kernel<64,8,4>` (per launch = 2 048 evaluations) |\n|---|---:|---:|\n")
        for k in sorted(set(p) | set(h)):
            g.write(f"| {k} | {p.get(k, float('nan')):.5g} | {h.get(k, float('nan')):.5g} |\n")
        evals_p = 4096 * 201
        dense = 33.01e6 / (2 * 16 * 16 * 32)          # dense-count MFMAs per evaluation (16x16x32 fp16)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "GRBM_GUI_ACTIVE" in p:
            simd_cycles = p["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0      # shader cycles of the launch x 1 024 SIMDs (as in r02)
            g.write("\nDerived for the persistent kernel (same reading as `r02_pmc_sq_net_forward.md`): the launch lasts "
                    f"GRBM_GUI_ACTIVE / 8 = {p['GRBM_GUI_ACTIVE'] / 8.0:.4g} shader cycles on 1 024 SIMDs; the matrix pipes are "
                    f"busy SQ_VALU_MFMA_BUSY_CYCLES / that = **{p['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f}** of the "
                    f"SIMD-cycles (= SQ_INSTS_MFMA x 16: {p.get('SQ_INSTS_MFMA', 0) * 16:.4g}); MFMAs issued per evaluation "
                    f"{p.get('SQ_INSTS_MFMA', 0) / evals_p:.0f} of {dense:.0f} dense (out-of-board tile-taps skipped); "
                    f"VALU (incl. MFMA) / SALU wave-instructions per evaluation {p.get('SQ_INSTS_VALU', 0) / evals_p:.0f} / "
                    f"{p.get('SQ_INSTS_SALU', 0) / evals_p:.0f} -- the tree step's share is what the stand-alone network kernel "
                    "does not have: it issues 1 703 MFMAs + ~890 other VALU + ~150 SALU per evaluation (r02 counters of the "
                    "`<64,16,8>` shape scaled), the persistent kernel 3 850 other VALU + 1 226 SALU.  Wave cycles: "
                    f"{p.get('SQ_ACTIVE_INST_ANY', 0) / max(p.get('SQ_WAVE_CYCLES', 1), 1):.2f} issuing, "
                    f"{p.get('SQ_WAIT_INST_ANY', 0) / max(p.get('SQ_WAVE_CYCLES', 1), 1):.2f} issue-stalled, "
                    f"{p.get('SQ_WAIT_ANY', 0) / max(p.get('SQ_WAVE_CYCLES', 1), 1):.2f} waiting (s_waitcnt / barriers).\n")


if __name__ == "__main__":
    os.makedirs(out, exist_ok=True)
    kernel_stats()
    sq_table()
    print("ok")
