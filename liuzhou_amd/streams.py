"""HIP streams that really run side by side.

The runtime multiplexes streams onto a few hardware queues (4 by default) and picks the queue of a new stream from the
pool's reference counts, so two streams created back to back CAN share a queue depending on what the process created and
destroyed before -- their kernels then run strictly one after the other.  Measured in round 5
(`scripts/micro/stream_queues.py`, `profiles/r05_experiments.md`): with one extra stream alive the next pair was
serialised; inside a long-lived process (bench.py's later legs, a worker after an earlier engine) the two halves of a C2
search then took 31 ms per ply instead of 20.5.  `overlapping_streams` therefore PROBES: two spin kernels of ~0.3 ms, one
per stream, take ~0.3 ms together if the queues differ and ~0.6 ms if not; candidates that collide are held (so that
the pool moves on) until a set that overlaps pairwise is found, then released."""
from __future__ import annotations

import os
import time
from typing import List, Tuple

import torch

_SPIN_CYCLES = [0]


def _spin_cycles(dev: torch.device) -> int:
    """Cycle count of `torch.cuda._sleep` worth ~0.3 ms on this device (calibrated once)."""
    if _SPIN_CYCLES[0] <= 0:
        n = 200_000
        torch.cuda._sleep(n)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        torch.cuda._sleep(n)
        torch.cuda.synchronize(dev)
        dt = max(1e-6, time.perf_counter() - t0)
        _SPIN_CYCLES[0] = max(10_000, min(50_000_000, int(n * 3e-4 / dt)))
    return _SPIN_CYCLES[0]


def _overlap(dev: torch.device, s1: torch.cuda.Stream, s2: torch.cuda.Stream) -> bool:
    n = _spin_cycles(dev)
    for s in (s1, s2):                                           # first use of a stream: set-up costs, not a sample
        with torch.cuda.stream(s):
            torch.cuda._sleep(n)
    ratios = []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(s1):
            torch.cuda._sleep(n)
        torch.cuda.synchronize(dev)
        one = time.perf_counter() - t0
        t0 = time.perf_counter()
        with torch.cuda.stream(s1):
            torch.cuda._sleep(n)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(n)
        torch.cuda.synchronize(dev)
        both = time.perf_counter() - t0
        ratios.append(both / max(one, 1e-9))
    return sorted(ratios)[1] < 1.5                               # median: ~1.05 on two queues, ~1.95 on one


def stream_pair_mode() -> str:
    """LZ_STREAM_PAIR: "probe" (default) = an equal-priority pair, probed when drawn; "priority" (round 6) = the k streams
    get DISTINCT priorities (-1, 0[, 1]), which the runtime serves from distinct hardware-queue pools -- the mapping is then
    a property of how the streams were created, not of what else the process has alive.  Same-box A/B inside bench.py at C2
    (profiles/r06_experiments.md section 2): 196.4 / 196.6 k positions/s against 196.8 / 196.3 k for the probed pair.  But
    inside bench.py's default sequence the priority pair ran the whole C2 runner leg at 31 ms per ply instead of 20.5, every
    time: both halves side by side in time and no faster than one after the other.  Distinct queues are not a guarantee;
    the symmetric pair stays the default, and DualStreamTreeMCTS judges every run against a reference search with both
    halves on one stream and replaces a pair that does not share the chip by a freshly probed equal-priority one."""
    return os.environ.get("LZ_STREAM_PAIR", "probe").strip().lower()


def overlapping_streams(device, k: int = 2, max_tries: int = 12, mode: str = None) -> Tuple[torch.cuda.Stream, ...]:
    """`k` streams on `device` whose kernels overlap pairwise (see the module docstring).  Falls back to the last
    candidates after `max_tries` collisions (results never depend on the overlap, only the speed does).
    `mode`: "priority" / "probe" (default: `stream_pair_mode()`)."""
    dev = torch.device(device)
    chosen: List[torch.cuda.Stream] = []
    rejected: List[torch.cuda.Stream] = []
    with torch.cuda.device(dev):
        _spin_cycles(dev)
        if (mode or stream_pair_mode()) == "priority" and int(k) <= 3:
            # priorities on this runtime: -1 (high), 0 (normal), 1 (low) where the range allows; k = 2 -> (high, normal)
            prios = [-1, 0, 1][: int(k)]
            cand = [torch.cuda.Stream(dev, priority=p) for p in prios]
            if all(_overlap(dev, a, b) for i, a in enumerate(cand) for b in cand[i + 1:]):
                overlapping_streams.last_rejected = 0
                overlapping_streams.last_mode = "priority"
                return tuple(cand)
            rejected.extend(cand)                                # not expected; fall through to the probed draw
        overlapping_streams.last_mode = "probe"
        while len(chosen) < int(k):
            cand = torch.cuda.Stream(dev)
            if all(_overlap(dev, c, cand) for c in chosen) or len(rejected) >= int(max_tries):
                chosen.append(cand)
            else:
                rejected.append(cand)                            # keep it alive: the next candidate gets another queue
    overlapping_streams.last_rejected = len(rejected)
    del rejected
    return tuple(chosen)


overlapping_streams.last_rejected = 0
overlapping_streams.last_mode = "probe"


# Stream capture mode of every hipGraph this package records.  "thread_local": only the CAPTURING thread is held to the
# capture rules.  The default ("global") lets an allocation or a pinned-memory call made by ANY thread invalidate a capture
# in progress -- and the streaming worker has a copier thread that pins staging buffers and copies segments while the
# playing thread may be capturing a launch form it meets for the first time (the compact lists, in the drain of a wave).
CAPTURE_MODE = "thread_local"
