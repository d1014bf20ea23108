"""Picklable stand-in for the GPU worker in the self-play stage tests: it goes through the real chunk writer
(`write_worker_chunks`) with seeded random trajectories instead of searching on a device."""
import torch

from liuzhou_amd.self_play_types import SelfPlayV1Stats
from liuzhou_amd.self_play_worker import write_worker_chunks
from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch


def random_batch(n: int, seed: int) -> TensorSelfPlayBatch:
    g = torch.Generator().manual_seed(int(seed))
    legal = torch.rand((n, 220), generator=g) < 0.15
    legal[:, 0] = True
    pol = torch.rand((n, 220), generator=g) * legal
    pol = pol / pol.sum(1, keepdim=True)
    return TensorSelfPlayBatch(
        state_tensors=(torch.rand((n, 11, 6, 6), generator=g) < 0.2).to(torch.float32), legal_masks=legal,
        policy_targets=pol.to(torch.float32),
        value_targets=torch.randint(-1, 2, (n,), generator=g).to(torch.float32),
        soft_value_targets=torch.rand((n,), generator=g) * 2 - 1)


def stub_worker(**kw):
    calls = {"n": 0}

    def run_once(games):
        calls["n"] += 1
        n = games * 7
        batch = random_batch(n, kw["seed"] * 131 + calls["n"])
        keys = ("root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms")
        st = SelfPlayV1Stats(num_games=games, num_positions=n, black_wins=games // 3, white_wins=games // 4,
                             draws=games - games // 3 - games // 4, avg_game_length=7.0, elapsed_sec=0.5,
                             positions_per_sec=n / 0.5, games_per_sec=games / 0.5,
                             step_timing_ms={k: 1.0 for k in keys}, step_timing_ratio={k: 0.25 for k in keys},
                             step_timing_calls={k: 1 for k in keys}, mcts_counters={"leaf_eval_count": n * 3},
                             piece_delta_buckets={"0": games}, device=kw["shard_device"])
        return batch, st

    games = int(kw["shard_games"])
    return write_worker_chunks(
        run_once, worker_idx=kw["worker_idx"], device=kw["shard_device"], games=games,
        games_per_chunk=max(1, min(games, int(kw["concurrent_games_per_device"]))),
        soft_label_alpha=kw["soft_label_alpha"], chunk_dir=kw["chunk_output_dir"], chunk_prefix=kw["chunk_file_prefix"],
        chunk_file_ext=kw["chunk_file_ext"], output_path=kw["output_path"],
        target_samples_per_shard=kw["target_samples_per_shard"], chunk_target_bytes=kw["chunk_target_bytes"],
        meta_common={"search_backend": kw["search_backend"], "opening_random_moves": kw["opening_random_moves"]})


def stub_stream_worker(**kw):
    """The STREAMED shard writer (`StreamedShardFiles`: what `stream_worker_shard` feeds from the finished-row log) with
    the same seeded trajectories, cut into segments of uneven size and written by two threads out of order -- on the CPU.
    The staging tensors are larger than a segment, as the pinned staging buffers of the worker are."""
    import threading
    from liuzhou_amd.self_play_worker import StreamedShardFiles, merge_self_play_stats
    games = int(kw["shard_games"])
    n = games * 7
    # the rows the chunk loop of `stub_worker` produces for this shard (one seeded batch per chunk of games)
    per, parts, left, call = max(1, min(games, int(kw["concurrent_games_per_device"]))), [], games, 0
    while left > 0:
        call += 1
        parts.append(random_batch(min(per, left) * 7, kw["seed"] * 131 + call))
        left -= min(per, left)
    batch = TensorSelfPlayBatch(*(torch.cat([getattr(p, f) for p in parts]) for f in (
        "state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")))
    shard = StreamedShardFiles(
        device=kw["shard_device"], worker_idx=kw["worker_idx"], games=games,
        games_per_chunk=max(1, min(games, int(kw["concurrent_games_per_device"]))),
        soft_label_alpha=kw["soft_label_alpha"], chunk_dir=kw["chunk_output_dir"], chunk_prefix=kw["chunk_file_prefix"],
        chunk_file_ext=kw["chunk_file_ext"], output_path=kw["output_path"],
        target_samples_per_shard=kw["target_samples_per_shard"], chunk_target_bytes=kw["chunk_target_bytes"],
        meta_common={"search_backend": kw["search_backend"], "opening_random_moves": kw["opening_random_moves"]})
    cuts = [0]
    while cuts[-1] < n:
        cuts.append(min(n, cuts[-1] + 9 + 4 * (len(cuts) % 3)))
    jobs = []
    for number, (a, b) in enumerate(zip(cuts, cuts[1:])):
        rows = b - a
        staging = tuple(torch.cat([getattr(batch, f)[a:b], torch.zeros((5, *getattr(batch, f).shape[1:]),
                                                                       dtype=getattr(batch, f).dtype)])
                        for f in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"))
        jobs.append((staging, rows, shard.plan_segment(rows, number)))            # segment order, one thread
    halves = [jobs[1::2], jobs[0::2]]                                             # written out of order, two threads
    ths = [threading.Thread(target=lambda js=js: [shard.write_segment(*j) for j in js]) for js in halves]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    keys = ("root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms")
    st = SelfPlayV1Stats(num_games=games, num_positions=n, black_wins=games // 3, white_wins=games // 4,
                         draws=games - games // 3 - games // 4, avg_game_length=7.0, elapsed_sec=0.5,
                         positions_per_sec=n / 0.5, games_per_sec=games / 0.5, step_timing_ms={k: 1.0 for k in keys},
                         step_timing_ratio={k: 0.25 for k in keys}, step_timing_calls={k: 1 for k in keys},
                         mcts_counters={"leaf_eval_count": n * 3}, piece_delta_buckets={"0": games},
                         device=kw["shard_device"])
    return shard.finish(merge_self_play_stats([st], 0.5), len(jobs))
