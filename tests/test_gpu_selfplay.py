"""GPU: the root-PUCT self-play wave loop on the HIP operators reproduces the reference's own trace
(tests/golden/g8_selfplay.npz: v1 runner, 4 games x 32 sims, tiny net, deterministic picks)."""
import numpy as np
import pytest
import torch

from tests.golden_utils import load

pytestmark = pytest.mark.gpu


def test_gpu_root_selfplay_matches_reference_trace():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    z = load("g8_selfplay.npz")
    torch.manual_seed(7)
    model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to("cuda:0")
    batch, stats = self_play_v1_gpu(model, num_games=4, mcts_simulations=32, temperature_init=1.0,
                                    temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                    device="cuda:0", add_dirichlet_noise=False, soft_value_k=2.0,
                                    opening_random_moves=0, max_game_plies=512, sample_moves=False,
                                    concurrent_games=4, autocast_dtype="float32", collect_step_timing=True)
    n = int(z["num_positions"])
    assert stats.num_positions == n
    want_states = np.unpackbits(z["state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert np.array_equal(batch.state_tensors.cpu().numpy(), want_states)
    assert np.array_equal(batch.legal_masks.cpu().numpy(), np.unpackbits(z["legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(batch.policy_targets.cpu().numpy(), z["policy_targets"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(batch.value_targets.cpu().numpy(), z["value_targets"])
    np.testing.assert_allclose(batch.soft_value_targets.cpu().numpy(), z["soft_value_targets"], atol=1e-6, rtol=0)
    assert (stats.black_wins, stats.white_wins, stats.draws) == (int(z["black_wins"]), int(z["white_wins"]), int(z["draws"]))
    assert abs(stats.avg_game_length - float(z["avg_game_length"])) < 1e-6
    assert set(stats.step_timing_ms) == {"root_puct_ms", "pack_writeback_ms", "self_play_step_ms", "finalize_ms"}


def test_gpu_selfplay_sampled_contract():
    """Contract asserts of tests/v1/test_v1_tensor_pipeline_smoke.py:73-111 (noise + sampling + opening moves)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    torch.manual_seed(0)
    model = ChessNet().eval().to("cuda:0")
    samples, stats = self_play_v1_gpu(model=model, num_games=3, mcts_simulations=4, temperature_init=1.0,
                                      temperature_final=0.1, temperature_threshold=4, exploration_weight=1.0,
                                      device="cuda:0", add_dirichlet_noise=True, dirichlet_alpha=0.3,
                                      dirichlet_epsilon=0.25, soft_value_k=2.0, opening_random_moves=2,
                                      max_game_plies=96, sample_moves=True, concurrent_games=2)
    assert stats.num_games == 3 and samples.num_samples > 0
    assert samples.state_tensors.shape[1:] == (11, 6, 6)
    assert samples.legal_masks.shape == (samples.num_samples, 220)
    assert samples.policy_targets.shape == (samples.num_samples, 220)
    assert samples.value_targets.shape == (samples.num_samples,)
    assert int(stats.mcts_counters.get("forced_uniform_pick_count", 0)) > 0
    pol = samples.policy_targets
    assert torch.allclose(pol.sum(1), torch.ones_like(pol.sum(1)), atol=1e-4)
    assert bool((pol[~samples.legal_masks] == 0).all())
    assert bool(torch.isfinite(samples.value_targets).all()) and bool(torch.isfinite(samples.soft_value_targets).all())
    assert bool((samples.value_targets.abs() <= 1).all())


def test_tree_self_play_runner_dual_stream_matches_contract():
    """self_play_tree_gpu with the two-stream search: same tensor contract, every game played to the ply cap."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import self_play_tree_gpu
    torch.manual_seed(2)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to("cuda:0"))
    for dual in (True, False):
        batch, stats = self_play_tree_gpu(net, num_games=65, mcts_simulations=12, temperature_init=1.0,
                                          temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                          device="cuda:0", max_game_plies=30, concurrent_games=65, dual_stream=dual)
        assert batch.num_samples == 65 * 30 == stats.num_positions
        assert torch.allclose(batch.policy_targets.sum(1), torch.ones(batch.num_samples, device="cuda:0"), atol=1e-4)
        assert bool((batch.policy_targets[~batch.legal_masks] == 0).all())
        assert bool(torch.isfinite(batch.value_targets).all())
        assert stats.mcts_counters["leaf_eval_count"] == 65 * 13 * 30
