"""Persistent search kernel (csrc/lz_search.hip) against the per-simulation launch pair, same box, same workload.

    python scripts/exp/exp_persistent.py [games] [sims] [steps] [model]

The `stagger:mode` settings (timing modes that skip / replace the tree step) need an experiment build of the library:
    hipcc ... -DLZ_EXP_SEARCH_MODES -o /tmp/libexp.so ...   and   LZ_HIP_LIB=/tmp/libexp.so python scripts/exp/exp_persistent.py
(the shipped library ignores LZ_EXP_SEARCH_MODE).
For each setting (LZ_TREE_PERSISTENT off = two streams of per-simulation launches; on with several staggers) the
steady-state self-play harness runs `steps` timed steps after a 2 s soak; with the persistent kernel the in-kernel
phase clocks (100 MHz ticks per workgroup in network passes / tree steps) of the last search are printed too."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay

games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 200
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
name = sys.argv[4] if len(sys.argv) > 4 else "b6c64"
settings = os.environ.get("LZ_EXP_SETTINGS", "off,0,20,40,60").split(",")
dev = torch.device("cuda:0")
torch.manual_seed(20260314)
model = ChessNet(**MODEL_CONFIGS[name]).eval().to(dev)


def run(persistent: bool, stagger: int, ticks: bool, mode: int = 0):
    os.environ["LZ_EXP_SEARCH_MODE"] = str(mode)
    os.environ["LZ_TREE_PERSISTENT"] = "1" if persistent else "0"
    os.environ["LZ_TREE_STAGGER_US"] = str(stagger)
    torch.manual_seed(9973)
    pop = SteadyStateTreeSelfPlay(FusedNet(model), games, sims=sims, device=dev, seed=9973, reuse_tree=True,
                                  reuse_factor=8.0, dual_stream=True, arena_rows=games * (steps + 160))
    eng = pop.mcts.engine if hasattr(pop.mcts, "engine") else None
    pt = eng.enable_phase_ticks() if (ticks and eng is not None and persistent) else None
    pop.preroll(120)
    pop.prepare()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        pop.step(); torch.cuda.synchronize(dev); n += 1
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        pop.step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    out = {"persistent": persistent, "stagger_us": stagger, "exp_mode": mode, "ms_per_step": round(dt * 1e3, 3),
           "positions_per_s": round(games / dt, 1), "us_per_sim": round(dt * 1e6 / (sims + 1), 2)}
    if pt is not None:
        t = pt.double().cpu()
        out["phase_us_per_sim"] = {"net_mean": round(float(t[:, 0].mean()) / 100 / (sims + 1), 2),
                                   "tree_mean": round(float(t[:, 1].mean()) / 100 / (sims + 1), 2),
                                   "net_max": round(float(t[:, 0].max()) / 100 / (sims + 1), 2),
                                   "tree_max": round(float(t[:, 1].max()) / 100 / (sims + 1), 2),
                                   "wg_total_max": round(float(t[:, :2].sum(1).max()) / 100 / (sims + 1), 2),
                                   "wg_total_mean": round(float(t[:, :2].sum(1).mean()) / 100 / (sims + 1), 2),
                                   "second_slot_wgs": int((t[:, 2].long() & 1).sum()), "distinct_cus": int(t[:, 3].unique().numel()),
                                   "max_wgs_per_cu": int(torch.bincount(t[:, 3].long()).max())}
        dump = os.environ.get("LZ_EXP_DUMP_TICKS")
        if dump:                                              # raw per-workgroup rows: net ticks, tree ticks, slot, CU key
            import numpy as np
            np.save(f"{dump}_s{stagger}_m{mode}.npy", pt.cpu().numpy())
    del pop
    torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


for s in settings:
    if s == "off":
        run(False, 0, False)
    elif ":" in s:                                   # stagger:mode  (timing experiments with wrong results)
        run(True, int(s.split(":")[0]), True, int(s.split(":")[1]))
    else:
        run(True, int(s), True)
