"""Do two HIP streams created back to back always land on different hardware queues?  (Round 5: the two halves of a C2
search ran one after the other in SOME worker runs.)  Two streams each run a chain of short kernels on a small grid; if
they overlap, both chains take about as long as one."""
import sys, time
import torch
dev = torch.device("cuda:0")
a = torch.zeros(1 << 16, device=dev); b = torch.zeros(1 << 16, device=dev)

def chain(t, n=3000):
    for _ in range(n):
        t.add_(1.0)

def measure(s1, s2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): chain(a)
    torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): chain(a)
    with torch.cuda.stream(s2): chain(b)
    torch.cuda.synchronize(); both = time.perf_counter() - t0
    return one, both

def graph_measure(s1, s2):
    """Same with captured graphs (what the engine replays): launch cost is off the host."""
    gs = []
    for s, t in ((s1, a), (s2, b)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            chain(t, 10); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                chain(t, 2000)
        gs.append(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): gs[0].replay()
    torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): gs[0].replay()
    with torch.cuda.stream(s2): gs[1].replay()
    torch.cuda.synchronize(); both = time.perf_counter() - t0
    return one, both

keep = []
for trial in range(12):
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    one, both = graph_measure(s1, s2)
    print(f"trial {trial:2d}: streams alive {len(keep):2d}  one chain {one*1e3:6.1f} ms, two chains on two streams {both*1e3:6.1f} ms  "
          f"ratio {both/one:.2f}  {'OVERLAP' if both < 1.5 * one else 'SERIALIZED'}", flush=True)
    # perturb the pool: keep an odd number of extra streams alive, drop the pair
    keep.append(torch.cuda.Stream(dev))
    if trial % 3 == 2:
        keep.pop(0)
    del s1, s2

# the fix: liuzhou_amd.streams.overlapping_streams probes and re-draws
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from liuzhou_amd.streams import overlapping_streams
keep = []
for trial in range(8):
    s1, s2 = overlapping_streams(dev, 2)
    one, both = graph_measure(s1, s2)
    print(f"probed pair {trial}: streams alive {len(keep):2d}  rejected candidates {overlapping_streams.last_rejected}  ratio {both/one:.2f}  "
          f"{'OVERLAP' if both < 1.5 * one else 'SERIALIZED'}", flush=True)
    keep.append(torch.cuda.Stream(dev))
    del s1, s2
