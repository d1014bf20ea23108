"""GPU: the tree worker plays different games in every chunk and reproduces itself for a given seed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tree_worker_chunks_differ_and_runs_repeat(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.self_play_worker import run_self_play_worker
    m = ChessNet(**MODEL_CONFIGS["b6c64"])
    stable_resnet_init(m, 20260314)
    ck = tmp_path / "model_state_cpu.pt"
    torch.save(m.state_dict(), ck)

    def run(tag, seed):
        out = tmp_path / f"{tag}.pt"
        run_self_play_worker(worker_idx=0, shard_device="cuda:0", shard_games=16, seed=seed, model_state_path=str(ck),
                             output_path=str(out), mcts_simulations=8, temperature_init=1.0, temperature_final=0.1,
                             temperature_threshold=10, exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
                             soft_value_k=2.0, opening_random_moves=0, max_game_plies=12, concurrent_games_per_device=8,
                             chunk_output_dir=str(tmp_path), chunk_file_prefix=tag, search_backend="portable")
        man = torch.load(out, weights_only=False)
        assert man["metadata"]["graph_retry_off"] is False and man["num_samples"] == 16 * 12
        st = man["stats"]
        assert st["step_timing_ms"]["root_puct_ms"] > 0 and st["step_timing_calls"]["root_puct_ms"] > 0
        return [torch.load(tmp_path / f, weights_only=False) for f in man["shard_files"]]

    a = run("a", 5)
    b = run("b", 5)
    c = run("c", 6)
    assert len(a) == 2                                       # 16 games through 8 slots: two chunks
    cat = lambda chunks, k: torch.cat([ch[k] for ch in chunks])
    assert not torch.equal(a[0]["policy_targets"], a[1]["policy_targets"])        # the second chunk plays other games
    assert torch.equal(cat(a, "policy_targets"), cat(b, "policy_targets"))        # same seed: the same games again
    assert torch.equal(cat(a, "state_tensors"), cat(b, "state_tensors"))
    assert not torch.equal(cat(a, "policy_targets"), cat(c, "policy_targets"))    # another seed: other games


def test_root_worker_seeds_give_different_games_and_repeat_with_opening_moves(tmp_path, monkeypatch):
    """Variant R through the worker with the reference script's default `opening_random_moves = 6`
    (scripts/big_train_v1.sh:42): the fused root search + device tail + streaming carry it (round 5; before, the option
    fell back to the operator chain), the Philox key comes from the worker's seed (before: one fixed key for every worker
    and chunk), the opening moves are legal and spread over the board."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.self_play_worker import run_self_play_worker
    m = ChessNet(**MODEL_CONFIGS["b6c64"])
    stable_resnet_init(m, 20260314)
    ck = tmp_path / "model_state_cpu.pt"
    torch.save(m.state_dict(), ck)

    def run(tag, seed):
        out = tmp_path / f"{tag}.pt"
        run_self_play_worker(worker_idx=0, shard_device="cuda:0", shard_games=256, seed=seed, model_state_path=str(ck),
                             output_path=str(out), mcts_simulations=16, temperature_init=1.0, temperature_final=0.1,
                             temperature_threshold=10, exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
                             soft_value_k=2.0, opening_random_moves=6, max_game_plies=16, concurrent_games_per_device=128,
                             chunk_output_dir=str(tmp_path), chunk_file_prefix=tag, search_backend="cuda_root")
        man = torch.load(out, weights_only=False)
        assert man["num_samples"] == 256 * 16 and man["metadata"]["opening_random_moves"] == 6
        assert man["stats"]["mcts_counters"].get("fused_root_search") == 1            # not the operator chain
        assert man["stats"]["mcts_counters"].get("stream_segments", 0) >= 1           # ... and streamed
        chunks = [torch.load(tmp_path / f, weights_only=False) for f in man["shard_files"]]
        cat = lambda k: torch.cat([c[k] for c in chunks])
        return cat("state_tensors"), cat("policy_targets"), cat("legal_masks")

    a, b, c = run("a", 5), run("b", 5), run("c", 6)
    key = lambda t: sorted(bytes(r) for r in t[0].reshape(t[0].shape[0], -1).to(torch.uint8).numpy())
    assert key(a) == key(b)                                   # same seed: the same positions (as a multiset of rows)
    assert key(a) != key(c)                                   # another seed: other games
    for st, pol, legal in (a, c):
        assert torch.allclose(pol.sum(1), torch.ones(pol.shape[0]), atol=1e-4) and bool((pol[~legal] == 0).all())
    # ply-1 positions (exactly one stone on the board): the opening move was drawn uniformly from 36 cells
    own_or_opp = a[0][:, 0] + a[0][:, 1]
    ply1 = own_or_opp.flatten(1).sum(1) == 1
    cells = own_or_opp[ply1].flatten(1).argmax(1)
    assert int(ply1.sum()) == 256 and len(set(cells.tolist())) >= 30          # 256 uniform draws over 36 cells
