"""Summaries of scripts/exp/prof_r05_tree.sh: per kernel, mean counter values per launch and the derived ratios."""
import collections, csv, glob, os, sys
out = sys.argv[1]
KEEP = ("tree_expand_select_kernel", "tree_advance_kernel", "tree_expand_kernel", "net_forward_kernel", "tree_finish_kernel")
def short(k):
    for n in KEEP:
        if n in k:
            t = k[k.index(n):]
            return t[:t.index("(")] if "(" in t else t[:60]
    return None
for tag_dir in sorted(glob.glob(os.path.join(out, "stats_*/"))):
    tag = tag_dir.rstrip("/").split("stats_")[-1]
    print(f"## {tag}\n")
    # kernel-trace statistics
    for f in glob.glob(os.path.join(tag_dir, "*", "*kernel_stats.csv")):
        print("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|")
        for r in csv.DictReader(open(f)):
            n = short(r["Name"]) or r["Name"][:50]
            print(f"| {n} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
        print()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("pmc1_", "pmc2_", "pmc3_"):
        for f in glob.glob(os.path.join(out, p + tag, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        print(f"**{k}** (mean per launch over {len(next(iter(d.values())))} launches)\n")
        for c in sorted(m):
            print(f"- {c}: {m[c]:.4g}")
        w = m.get("SQ_WAVES")
        if w:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"):
                if c in m:
                    print(f"- {c} per wave: {m[c] / w:.1f}")
            if "SQ_WAVE_CYCLES" in m:
                print(f"- SQ_WAVE_CYCLES per wave: {m['SQ_WAVE_CYCLES'] / w:.0f}")
        if "SQ_WAVE_CYCLES" in m:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
                if c in m:
                    print(f"- {c} / SQ_WAVE_CYCLES: {m[c] / m['SQ_WAVE_CYCLES']:.3f}")
        print()
