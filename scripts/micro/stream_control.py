"""Control, don't probe, the two-stream overlap.

Part 1 -- queue mapping: for 0..7 other streams alive in the process, draw a pair (a) of equal priority, unprobed,
(b) of DISTINCT priorities (-1, 0), and measure the overlap of two graph chains of small kernels (ratio of both / one:
~1.0-1.4 on two hardware queues, ~1.95 on one).  Does the priority pair always overlap, whatever else is alive?

Part 2 -- what the priority asymmetry costs or buys end to end: the C2 steady-state search (4 096 games, 200 sims, 6x64,
two engines on two streams) on a probed equal-priority pair against a (-1, 0) pair, alternating, same process, same seed;
positions/s and the engine's own overlap-watch verdicts (serial searches seen, redraws).

usage: python scripts/micro/stream_control.py [part1] [part2]      (default: both; one JSON line per measurement)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

dev = torch.device("cuda:0")


def chain_graphs(s1, s2, n=1500):
    a, b = torch.zeros(1 << 16, device=dev), torch.zeros(1 << 16, device=dev)
    gs = []
    for s, t in ((s1, a), (s2, b)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            for _ in range(10):
                t.add_(1.0)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    t.add_(1.0)
        gs.append(g)
    return gs


def ratio(s1, s2, gs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        gs[0].replay()
    torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        gs[0].replay()
    with torch.cuda.stream(s2):
        gs[1].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / one


def part1():
    keep = []
    for alive in range(8):
        row = {"part": 1, "other_streams_alive": alive}
        for tag, mk in (("equal", lambda: (torch.cuda.Stream(dev), torch.cuda.Stream(dev))),
                        ("priority", lambda: (torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)))):
            s1, s2 = mk()
            gs = chain_graphs(s1, s2)
            row[tag] = sorted(round(ratio(s1, s2, gs), 2) for _ in range(5))[2]
            del gs, s1, s2
        print(json.dumps(row), flush=True)
        keep.append(torch.cuda.Stream(dev))
        with torch.cuda.stream(keep[-1]):
            torch.zeros(8, device=dev).add_(1.0)
    torch.cuda.synchronize()


def part2(steps=60, rounds=3):
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import SteadyStateTreeSelfPlay
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    pops = {}
    for mode in ("probe", "priority"):
        os.environ["LZ_STREAM_PAIR"] = mode
        pop = SteadyStateTreeSelfPlay(net, 4096, sims=200, device=dev, seed=9973, reuse_tree=True, dual_stream=True)
        pop.preroll(120)
        pop.prepare()
        for _ in range(8):
            pop.step()
        torch.cuda.synchronize()
        pops[mode] = pop
    for r in range(rounds):
        for mode, pop in pops.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                pop.step()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            m = pop.mcts
            print(json.dumps({"part": 2, "round": r, "pair": mode, "positions_per_s": round(4096 * steps / dt),
                              "ms_per_step": round(dt / steps * 1e3, 3), "stream_redraws": int(m.stream_redraws),
                              "priorities": [int(s.priority) for s in m.streams]}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["part1", "part2"]
    if "part1" in what:
        part1()
    if "part2" in what:
        part2()
