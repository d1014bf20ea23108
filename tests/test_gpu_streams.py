"""GPU: streams that really run side by side (liuzhou_amd/streams.py) and the two-stream search's overlap watch."""
import numpy as np
import pytest
import torch

from tests.golden_utils import load, states, FIELDS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_probed_stream_pairs_overlap():
    """Two new HIP streams can share a hardware queue (their kernels then run one after the other);
    `overlapping_streams` probes with spin kernels and re-draws.  Whatever else is alive in the process, a probed pair runs
    two equal spin kernels in about the time of one."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    from liuzhou_amd.streams import _spin_cycles, overlapping_streams
    dev = torch.device(DEV)
    keep = []
    n = _spin_cycles(dev) * 4
    for trial in range(6):
        s1, s2 = overlapping_streams(dev, 2)
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
            ratios.append((time.perf_counter() - t0) / one)
        assert sorted(ratios)[1] < 1.5, (trial, ratios)
        keep.append(torch.cuda.Stream(dev))                       # shift the pool: the next pair is drawn in another state


def test_two_stream_search_notices_serialised_halves_and_draws_new_streams():
    """Both halves forced onto ONE stream (what a shared hardware queue does to them): the overlap watch -- events around
    the halves of the first searches, read later without waiting -- sees the union of the two intervals equal their sum and
    replaces the streams; results are those of a single engine either way."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import DualStreamTreeMCTS, PortableTreeMCTS
    from tests.tree_parity import to_gpu_batch
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    st_all = states(load("g1_rules.npz"), "s")
    idx = np.random.default_rng(3).integers(0, st_all["board"].shape[0], 2048)
    batch = to_gpu_batch({f: np.ascontiguousarray(np.asarray(st_all[f])[idx]) for f in FIELDS}, DEV)
    kw = dict(exploration_weight=1.0, add_dirichlet_noise=True, sample_moves=True, seed=5)
    dual = DualStreamTreeMCTS(net, 2048, 48, DEV, **kw)
    one = torch.cuda.Stream(torch.device(DEV))
    dual.streams = (one, one)                                       # serialised, as on one hardware queue
    temps = torch.ones((2048,), device=DEV)
    outs = []
    for k in range(10):
        outs.append(dual.search_batch(batch, temperatures=temps).chosen_action_indices.clone())
        if k == 0:      # the search that captured the graphs is not a witness: its halves are serial on the HOST
            assert dual.use_graph and not dual._watch and dual._serial_seen == 0 and dual._watch_left == 6
        torch.cuda.synchronize()
    assert dual.stream_redraws >= 1 and dual.streams[0] is not dual.streams[1] and one not in dual.streams
    single = PortableTreeMCTS(net, 2048, 48, DEV, **kw)
    want = single.search_batch(batch, temperatures=temps).chosen_action_indices
    assert torch.equal(outs[0], want)                               # the first move of both (same per-game RNG keys)
