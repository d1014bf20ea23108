"""Follow-up: (a) is the serialisation stable over time for a given pair, (b) do a normal- and a high-priority stream
always overlap, (c) does a probe with graph replays predict what graph chains do."""
import sys, time
import torch
dev = torch.device("cuda:0")
a = torch.zeros(1 << 16, device=dev); b = torch.zeros(1 << 16, device=dev)
def chain(t, n):
    for _ in range(n): t.add_(1.0)
def graphs(s1, s2, n=2000):
    gs = []
    for s, t in ((s1, a), (s2, b)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            chain(t, 10); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s): chain(t, n)
        gs.append(g)
    return gs
def ratio(s1, s2, gs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): gs[0].replay()
    torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): gs[0].replay()
    with torch.cuda.stream(s2): gs[1].replay()
    torch.cuda.synchronize(); both = time.perf_counter() - t0
    return both / one
def sleep_ratio(s1, s2, n=600000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): torch.cuda._sleep(n)
    torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): torch.cuda._sleep(n)
    with torch.cuda.stream(s2): torch.cuda._sleep(n)
    torch.cuda.synchronize(); both = time.perf_counter() - t0
    return both / one, one
keep = []
print("--- same priority, repeated measurements of one pair")
for trial in range(10):
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    sr = [round(sleep_ratio(s1, s2)[0], 2) for _ in range(3)]
    gs = graphs(s1, s2)
    r = [round(ratio(s1, s2, gs), 2) for _ in range(4)]
    sr2 = [round(sleep_ratio(s1, s2)[0], 2) for _ in range(2)]
    print(f"trial {trial}: alive {len(keep)}  sleep-probe {sr}  graph chains {r}  sleep-probe after {sr2}", flush=True)
    keep.append(torch.cuda.Stream(dev)); del s1, s2, gs
print("--- normal + high priority")
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
for trial in range(10):
    s1, s2 = torch.cuda.Stream(dev, priority=0), torch.cuda.Stream(dev, priority=-1)
    gs = graphs(s1, s2)
    r = [round(ratio(s1, s2, gs), 2) for _ in range(4)]
    print(f"trial {trial}: alive {len(keep)}  graph chains {r}", flush=True)
    keep.append(torch.cuda.Stream(dev)); del s1, s2, gs
print("sleep one ms", sleep_ratio(torch.cuda.Stream(dev), torch.cuda.Stream(dev))[1] * 1e3)
