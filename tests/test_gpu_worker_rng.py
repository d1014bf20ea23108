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
