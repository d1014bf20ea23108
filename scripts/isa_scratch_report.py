#!/usr/bin/env python3
"""Where do the network kernels' scratch (spill) instructions sit?  Compiles csrc/lz_net.hip to gfx950 assembly and,
per kernel, lists every scratch_load / scratch_store with its position relative to the residual-block loop (the only
place where time is spent: 94 % of the C3 launch).  The block loop is the innermost loop that contains MFMAs and four
`s_barrier`s; it is found as the backward branch whose body holds most of the kernel's v_mfma instructions.

    python scripts/isa_scratch_report.py > profiles/r04_scratch_isa.md
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "liuzhou_amd", "csrc", "lz_net.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "lz_net.s")
    defs = os.environ.get("LZ_ISA_DEFS", "").split()          # e.g. LZ_ISA_DEFS="-DLZ_NET_APF=1" for an experiment build
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           *defs, "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()

starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if l.startswith("_ZN") and "net_forward_kernel" in l.split(":")[0]]
print("# Scratch instructions of the fused network kernels vs the residual-block loop (gfx950 ISA, round 4)\n")
print("`python scripts/isa_scratch_report.py`; compiler: hipcc -O3 --offload-arch=gfx950.\n")
print("| kernel | v_mfma (static) | block loop (asm lines) | MFMAs inside it | scratch ops inside the loop | scratch ops outside |")
print("|---|---:|---|---:|---:|---:|")
for k, (i0, name) in enumerate(starts):
    i1 = next(j for j in range(i0, len(text)) if "s_endpgm" in text[j])
    body = text[i0:i1]
    m = re.search(r"net_forward_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name)
    label_at = {l.split(":")[0]: j for j, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    mf = [j for j, l in enumerate(body) if "v_mfma" in l]
    sc = [j for j, l in enumerate(body) if "scratch_" in l]
    best = None
    for j, l in enumerate(body):
        mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in label_at and label_at[mm.group(1)] < j:
            a, b = label_at[mm.group(1)], j
            inside = sum(1 for x in mf if a <= x <= b)
            bars = sum(1 for x in range(a, b) if "s_barrier" in body[x])
            if bars >= 4 and (best is None or (b - a) < (best[1] - best[0])) and inside > 0.5 * len(mf):
                best = (a, b, inside)
    if best is None:
        print(f"| <{m.group(1)},{m.group(2)},{m.group(3)}> | {len(mf)} | not found | | | {len(sc)} |")
        continue
    a, b, inside = best
    sin = [x for x in sc if a <= x <= b]
    sout = [x for x in sc if not (a <= x <= b)]
    print(f"| <{m.group(1)},{m.group(2)},{m.group(3)}> | {len(mf)} | {a}-{b} of {len(body)} | {inside} | "
          f"{len(sin)} | {len(sout)} (asm lines {', '.join(str(x) for x in sout[:12])}{' ...' if len(sout) > 12 else ''}) |")
print("\nThe spilled values (the thread id, a few per-lane addresses of the staging / head phases, one f4 of per-channel "
      "parameters) are stored before the block loop and reloaded after it, once per pass of S samples; no scratch "
      "instruction executes inside the loop that holds the MFMAs.")
