"""CPU: host-side logic that needs no GPU -- C ABI exports, storage planning, stats merging, loud failure."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports every LZ_API function of include/liuzhou_hip.h."""
    from liuzhou_amd.build import build_hip
    lib = ctypes.CDLL(build_hip())
    header = open(os.path.join(ROOT, "include", "liuzhou_hip.h")).read()
    declared = set(re.findall(r"LZ_API\s+[\w\s\*]+?\b(lz_\w+)\s*\(", header))
    assert len(declared) >= 20
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in liuzhou_hip.h but not exported"
    from liuzhou_amd import _lib
    assert set(_lib.SYMBOLS) <= declared
    lib.lz_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.lz_version()


def test_engines_fail_loudly_without_gpu():
    """The v0_core OPERATORS dispatch CPU tensors to the host build (tests/test_host_ops.py); the search engines and the
    network kernel are HIP-only and say so."""
    from liuzhou_amd.tree_engine import TreeEngine
    with pytest.raises(RuntimeError, match="HIP device"):
        TreeEngine(4, 8, "cpu")
    from liuzhou_amd.game_rng import GameRng
    with pytest.raises(RuntimeError, match="HIP device"):
        GameRng(4, "cpu")


def test_storage_planning_and_payload(tmp_path):
    from liuzhou_amd.self_play_storage import plan_sample_ranges, save_self_play_payload, split_counts, estimate_bytes_per_sample
    from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
    assert split_counts(10, 3) == [4, 3, 3] and split_counts(0, 3) == [] and split_counts(2, 5) == [1, 1]
    assert plan_sample_ranges(total_samples=10, num_shards=1) == [(0, 10)]
    assert plan_sample_ranges(total_samples=10, num_shards=1, target_samples_per_shard=4) == [(0, 4), (4, 7), (7, 10)]
    n = 7
    b = TensorSelfPlayBatch(torch.zeros(n, 11, 6, 6), torch.zeros(n, 220, dtype=torch.bool), torch.zeros(n, 220),
                            torch.zeros(n), torch.zeros(n))
    assert estimate_bytes_per_sample(b) == 11 * 36 * 4 + 220 + 220 * 4 + 4 + 4          # 2692 B / sample
    assert len(plan_sample_ranges(total_samples=n, num_shards=1, chunk_target_bytes=1024, bytes_per_sample=2692)) == n
    p = tmp_path / "shard.pt"
    save_self_play_payload(path=str(p), samples=b, stats_payload={"a": 1}, metadata={"payload_format": "v1_sharded_shard"})
    got = torch.load(p)
    assert set(got) == {"state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets", "stats", "metadata"}


def test_chunk_files_hold_only_their_own_rows(tmp_path):
    """A chunk sliced out of a host batch must not drag the whole batch's storage into its file (torch.save writes a
    tensor's entire underlying storage): file size ~ rows x 2692 B, and the loaded tensors own just their bytes."""
    from liuzhou_amd.self_play_storage import save_self_play_payload, slice_batch_cpu
    from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
    n = 2000
    b = TensorSelfPlayBatch(torch.rand(n, 11, 6, 6), torch.rand(n, 220) < 0.1, torch.rand(n, 220), torch.rand(n), torch.rand(n))
    part = slice_batch_cpu(b, start=100, end=300)
    assert part.state_tensors.untyped_storage().nbytes() == 200 * 11 * 36 * 4
    p = tmp_path / "chunk.pt"
    save_self_play_payload(path=str(p), samples=part, stats_payload={}, metadata={})
    assert 200 * 2692 <= os.path.getsize(p) < 200 * 2692 + 16384
    q = tmp_path / "view.pt"                                      # views handed straight to the writer are materialised too
    view = TensorSelfPlayBatch(*(getattr(b, f)[100:300] for f in ("state_tensors", "legal_masks", "policy_targets",
                                                                   "value_targets", "soft_value_targets")))
    save_self_play_payload(path=str(q), samples=view, stats_payload={}, metadata={})
    assert os.path.getsize(q) < 200 * 2692 + 16384
    got = torch.load(q)
    assert torch.equal(got["policy_targets"], b.policy_targets[100:300])


def test_target_summary_and_stats_merge():
    from liuzhou_amd.self_play_worker import merge_self_play_stats, merge_target_summaries, summarize_scalar_targets
    from liuzhou_amd.self_play_types import SelfPlayV1Stats
    s = summarize_scalar_targets(torch.tensor([1.0, -1.0, 0.0, float("nan"), 0.07]))
    assert (s["total"], s["finite_count"], s["nonfinite_count"], s["positive_count"], s["negative_count"], s["zero_count"]) == (5, 4, 1, 2, 1, 1)
    assert s["ge_abs_0p05_count"] == 3 and s["ge_abs_0p10_count"] == 2
    m = merge_target_summaries([s, s])
    assert m["total"] == 10 and abs(m["abs_mean"] - s["abs_mean"]) < 1e-12
    mk = lambda g, p: SelfPlayV1Stats(g, p, 1, 0, g - 1, 100.0, 2.0, p / 2.0, g / 2.0, {"root_puct_ms": 5.0}, {}, {"root_puct_ms": 2},
                                      {"leaf_eval_count": 10}, {str(d): (1 if d == 0 else 0) for d in range(-18, 19)}, device="cuda:0")
    merged = merge_self_play_stats([mk(2, 200), mk(4, 400)], elapsed_sec=3.0)
    assert merged.num_games == 6 and merged.num_positions == 600 and merged.positions_per_sec == 200.0
    assert merged.mcts_counters["leaf_eval_count"] == 20 and merged.piece_delta_buckets["0"] == 2
    d = merged.to_dict()
    assert set(d["step_timing_ms"]) >= {"root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms"}


def test_v1_helper_truth_tables():
    """tests/v1/test_v1_tensor_pipeline_smoke.py:20-70 (soft tanh, terminal mask, perspective flip) -- pure torch helpers."""
    import math
    from liuzhou_amd.mcts_gpu import GpuStateBatch, V1RootMCTS
    board = torch.zeros((3, 6, 6), dtype=torch.int8)
    board[0, :2, :] = 1
    board[1, :2, :] = -1
    soft = V1RootMCTS._soft_tanh_from_board_black(board, soft_value_k=2.0)
    assert float(soft[0]) == pytest.approx(math.tanh(4.0 / 3.0)) and float(soft[0]) > float(soft[2]) > float(soft[1])
    board = torch.zeros((3, 6, 6), dtype=torch.int8)
    board[0, 0, 0] = 1; board[1, 0, 0] = 1; board[1, 0, 1] = -1; board[2, 0, 0] = 1; board[2, 0, 1] = -1
    z = torch.zeros((3,), dtype=torch.int64)
    batch = GpuStateBatch(board, torch.zeros((3, 6, 6), dtype=torch.bool), torch.zeros((3, 6, 6), dtype=torch.bool),
                          torch.tensor([2, 4, 1]), torch.ones(3, dtype=torch.int64), z.clone(), z.clone(), z.clone(), z.clone(),
                          z.clone(), torch.tensor([0, 144, 0]), torch.tensor([0, 0, 36]))
    assert V1RootMCTS._terminal_mask_from_next_state(batch).tolist() == [False, True, True]
    aligned = V1RootMCTS._child_values_to_parent_perspective(torch.tensor([0.2, -0.5, 0.8, -0.1]), torch.tensor([1, 1, -1, -1]),
                                                             torch.tensor([1, -1, -1, 1]))
    assert torch.allclose(aligned, torch.tensor([0.2, 0.5, 0.8, 0.1]), atol=1e-6)


def test_trajectory_buffer_device_cursor_bookkeeping():
    """Host side of the device-side appends (wave_tail.WaveTail): the row cursor is a tensor the device advances;
    the host tracks an upper bound, reads it back only for growth / host appends / build, and growth keeps the rows."""
    import torch
    from liuzhou_amd.trajectory_buffer import TensorTrajectoryBuffer
    buf = TensorTrajectoryBuffer("cpu", 220, max_steps_hint=1, concurrent_games_hint=1, initial_capacity=8)
    cur = buf.reserve_rows(3)
    assert int(cur.item()) == 0 and buf.capacity == 8
    st, lg, pol, val, soft, sign = buf.arena()
    st[:3] = 1.0; pol[:3] = 0.5; lg[:3] = True; sign[:3] = 1; val[:3] = 0.25; soft[:3] = 0.5
    cur.add_(3)                                            # what lz_wave_record does on the device
    cur2 = buf.reserve_rows(3)                             # upper bound 6 <= 8: no read-back, same arena
    assert cur2 is cur and buf.arena()[0] is st
    cur.add_(2)                                            # only two of the three slots were live
    buf.reserve_rows(4)                                    # upper bound 10 > 8 -> read back 5, 5 + 4 > 8 -> grow
    assert buf.capacity >= 9 and buf.sync_cursor() == 5
    assert bool((buf.arena()[0][:3] == 1.0).all())          # rows survived the growth
    # a host-side append goes behind the device's rows and moves the cursor with it
    rows = buf.append_steps(torch.zeros(2, 11, 6, 6), torch.zeros(2, 220, dtype=torch.bool), torch.zeros(2, 220),
                            torch.tensor([1, -1]))
    assert rows.tolist() == [5, 6] and int(cur.item()) == 7
    out = buf.build()
    assert out.num_samples == 7 and float(out.value_targets[0]) == 0.25


def test_bench_clock_sampler_reads_sysfs_in_process(tmp_path, monkeypatch):
    """the power / clock sampler of bench.py must not start child processes (a `rocm-smi` child runs an
    `env` hop under a profiler's preload).  It reads the device's hwmon nodes in-process, and is off under a profiler."""
    import importlib.util
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lz_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hw = tmp_path / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "power1_input").write_text("1341000000\n")          # microwatts
    (hw / "freq1_input").write_text("1919000000\n")           # Hz

    def no_children(*a, **k):
        raise AssertionError("the sampler started a child process")
    monkeypatch.setattr(subprocess, "run", no_children)
    monkeypatch.setattr(subprocess, "Popen", no_children)
    s = bench.ClockSampler("0000:00:00.0", period=0.01, sysfs_base=str(tmp_path))
    assert s.available and "power1_input" in s.source and "freq1_input" in s.source
    t0 = time.perf_counter()
    s.start()
    time.sleep(0.1)
    out = s.stop(t0, time.perf_counter())
    assert out["samples_in_timed_region"] >= 2 and out["failed_samples"] == 0
    assert out["power_w_mean"] == 1341.0 and out["sclk_mhz_mean"] == 1919.0
    # pp_dpm_sclk fallback (no freq1_input): the starred level
    (hw / "freq1_input").unlink()
    (tmp_path / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2100Mhz *\n")
    s2 = bench.ClockSampler("0000:00:00.0", period=0.01, sysfs_base=str(tmp_path))
    s2._once()
    assert s2.samples and s2.samples[-1][2] == 2100.0
    # nothing to read: unavailable, never raises
    s3 = bench.ClockSampler("0000:00:00.0", sysfs_base=str(tmp_path / "missing"))
    assert isinstance(s3.available, bool)
    for var in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES"):
        monkeypatch.setenv(var, "/opt/rocm/lib/librocprofiler-sdk-tool.so")
        assert bench._under_profiler()
        monkeypatch.delenv(var)
    assert not bench._under_profiler() or any("rocprof" in os.environ.get(k, "").lower() for k in os.environ)


def test_bench_also_summary_is_compact_and_last():
    """`also_summary` is the digest the driver's truncated record of the line must still contain: one row per workload,
    one number per runner leg, failures named, and it is the LAST key of the printed line."""
    import json
    import bench
    leg = lambda v, ms, f, p: {"value": v, "ms_per_step": ms, "roofline": {"frac": f, "kernel_probe": {"frac": p}},
                               "cpu_baseline": {"value": 1.5}}
    out = leg(10000.0, 1600.0, 0.69, 0.72)
    out["also"] = {"C2": leg(196000.0, 20.8, 0.52, 0.50), "C2_fp32": {"failed": "RuntimeError('x')"},
                   "R_b10c128": leg(580000.0, 28.0, 0.67, 0.71),
                   "runner": {"self_play_tree_gpu@C2": {"value": 177000.0}, "run_self_play_worker@R_b10c128": {"failed": "x"}}}
    s = bench.also_summary(out)
    assert s["headline"] == [10000.0, 1600.0, 0.69, 0.72] and s["C2"] == [196000.0, 20.8, 0.52, 0.50]
    assert s["C2_fp32"] == "failed" and s["R_b10c128"][0] == 580000.0
    assert s["runner"] == {"self_play_tree_gpu@C2": 177000.0, "run_self_play_worker@R_b10c128": "failed"}
    assert s["cpu"]["headline"] == 1.5 and s["cpu"]["C2"] == 1.5
    out["also_summary"] = s
    line = json.dumps(out)
    assert line.rstrip("}").endswith('probe"') or list(out)[-1] == "also_summary"
    assert len(json.dumps(s)) < 1200                              # fits the 2 000 characters the driver keeps


def test_edge_pool_sizing_arithmetic():
    """chunk_cap_for: a game's chunk list must hold the worst case of its node arena (72 children everywhere, a run never
    straddles a chunk, so up to 71 records of a chunk stay unused) -- the condition lz_tree_advance checks."""
    from liuzhou_amd.tree_engine import chunk_cap_for, EDGE_CHUNK, MAX_CHILDREN
    for node_cap in (6, 42, 1002, 8202, 32802, 65536):
        for chunk in (128, 1024, 4096):
            cap = chunk_cap_for(node_cap, chunk)
            assert cap * (chunk - (MAX_CHILDREN - 1)) >= node_cap * MAX_CHILDREN
            assert (cap - 2) * (chunk - (MAX_CHILDREN - 1)) < node_cap * MAX_CHILDREN      # and not wastefully long
    assert EDGE_CHUNK == 1024 and EDGE_CHUNK & (EDGE_CHUNK - 1) == 0


def test_every_script_compiles():
    """scripts/ holds the CLIs and measurement programs, scripts/exp the one-off experiment / profiling programs of
    earlier rounds (they only run on the GPU box): at least their syntax is
    checked here, so that a refactor of the package does not leave them unparseable."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "scripts", "*.py")) + glob.glob(os.path.join(root, "scripts", "micro", "*.py")) +
                   glob.glob(os.path.join(root, "scripts", "exp", "*.py")) +
                   glob.glob(os.path.join(root, "oracle", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")])
    assert len(files) > 40
    for f in files:
        compile(open(f).read(), f, "exec")                       # syntax only: nothing is written, nothing is run


def test_overlap_watch_flags_disjoint_and_nested_halves(monkeypatch):
    """DualStreamTreeMCTS._check_overlap on recorded (start, end) brackets of the two halves: halves that share the chip
    take about the same time; one after the other (one hardware queue) and one nested in the other (one chain starved by the
    other -- seen once in round 6, unnoticed by the union test alone) both count as serial, two in a row replace the pair by
    an equal-priority, probed pair."""
    from liuzhou_amd import streams as S
    from liuzhou_amd.tree_engine import DualStreamTreeMCTS

    class Ev:
        def __init__(self, t): self.t = t
        def query(self): return True
        def elapsed_time(self, other): return other.t - self.t

    drawn = []
    def fake_streams(device, k=2, max_tries=12, mode=None):
        drawn.append(mode)
        return (object(), object())
    fake_streams.last_mode = "probe"
    monkeypatch.setattr(S, "overlapping_streams", fake_streams)

    def engine():
        d = DualStreamTreeMCTS.__new__(DualStreamTreeMCTS)
        d._watch, d._watch_left, d._searches, d._serial_seen, d.stream_redraws = [], 6, 0, 0, 0
        d._t_serial, d.overlap_ratio = None, None
        d.parts, d.device, d.streams, d._pair_mode = [None, None], "cuda:0", ("a", "b"), "probe"
        return d

    def feed(d, s0, e0, s1, e1, calib=False):
        d._watch.append([Ev(s0), Ev(e0), Ev(s1), Ev(e1), calib])
        d._check_overlap()

    d = engine()
    for _ in range(4):
        feed(d, 0.0, 20.0, 0.2, 20.5)                     # healthy: both halves ~20 ms, side by side
    assert d.stream_redraws == 0 and d._serial_seen == 0
    feed(d, 0.0, 15.4, 15.5, 31.0); feed(d, 0.0, 15.4, 15.5, 31.0)      # one after the other
    assert d.stream_redraws == 1 and drawn == ["probe"] and d._pair_mode == "probe"
    d = engine()
    feed(d, 0.0, 15.4, 0.1, 31.0)                          # nested: the second half starved until the first is done
    assert d._serial_seen == 1 and d.stream_redraws == 0
    feed(d, 0.0, 20.0, 0.1, 20.3)                          # a healthy search in between resets the count
    assert d._serial_seen == 0
    feed(d, 0.0, 15.4, 0.1, 31.0); feed(d, 0.0, 15.0, 0.1, 30.0)
    assert d.stream_redraws == 1 and d._pair_mode == "probe"
    d = engine()
    for _ in range(6):
        feed(d, 0.0, 16.0, 0.3, 21.0)                      # uneven but sharing the chip (ratio 0.76): not a failure
    assert d.stream_redraws == 0 and d._serial_seen == 0
    # (c) the third signature (round 6, a (-1, 0) priority pair in bench.py's runner leg): both halves side by side in TIME,
    # equally long -- but as long as one half after the other.  Only the reference search of the run (both halves on one
    # stream on purpose) tells: union > 0.85 of it = no overlap.
    d = engine()
    feed(d, 0.0, 31.2, 0.0, 31.0)
    assert d._serial_seen == 0                                 # without a reference this looks healthy
    feed(d, 0.0, 15.4, 15.5, 31.0, calib=True)                 # the reference: 15.4 + 15.5 ms one after the other
    assert abs(d._t_serial - 30.9) < 1e-6 and d._serial_seen == 0 and d.stream_redraws == 0
    feed(d, 0.0, 20.5, 0.0, 20.4)
    assert d._serial_seen == 0 and abs(d.overlap_ratio - 20.5 / 30.9) < 1e-6
    feed(d, 0.0, 31.2, 0.0, 31.0); feed(d, 0.0, 31.0, 0.0, 31.3)
    assert d.stream_redraws == 1 and drawn[-1] == "probe"

