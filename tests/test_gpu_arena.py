"""GPU: evaluation arena (f3) -- device-resident matches between two agents."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_random_vs_random_plays_every_game_to_the_end():
    _need_gpu()
    from liuzhou_amd.eval_arena import RandomAgent, play_matches
    a = play_matches(RandomAgent(), RandomAgent(), 96, DEV, seed=3)
    assert a.total_games == 96 and a.wins + a.losses + a.draws == 96
    cb = a.color_breakdown
    assert cb["black"]["games"] == cb["white"]["games"] == 48
    assert cb["black"]["wins"] + cb["white"]["wins"] == a.wins
    assert abs(a.win_rate + a.loss_rate + a.draw_rate - 1.0) < 1e-9
    b = play_matches(RandomAgent(), RandomAgent(), 96, DEV, seed=3)
    assert (a.wins, a.losses, a.draws) == (b.wins, b.losses, b.draws)            # seeded
    # random play is decisive: pieces get captured until one side drops below four
    assert a.wins + a.losses > 0


def test_checkpoint_agent_vs_random_and_vs_previous(tmp_path):
    _need_gpu()
    from liuzhou_amd.eval_arena import evaluate_checkpoint
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    paths = []
    for seed in (20260314, 7):
        m = ChessNet(**MODEL_CONFIGS["b6c64"])
        stable_resnet_init(m, seed)
        p = tmp_path / f"model_{seed}.pt"
        torch.save({"model_state_dict": m.state_dict()}, p)
        paths.append(str(p))
    r = evaluate_checkpoint(paths[0], None, num_games=33, device=DEV, mcts_simulations=8, opening_random_moves=4,
                            max_game_plies=160, seed=1)
    assert r["name"] == "vs_random" and r["total_games"] == 32                    # odd counts round down to even
    assert r["wins"] + r["losses"] + r["draws"] == 32
    for key in ("win_rate", "loss_rate", "draw_rate", "color_breakdown", "seed"):
        assert key in r
    r2 = evaluate_checkpoint(paths[0], paths[1], num_games=16, device=DEV, mcts_simulations=8, sample_moves=True,
                             temperature=1.0, max_game_plies=120, seed=2)
    assert r2["name"] == "vs_previous" and r2["wins"] + r2["losses"] + r2["draws"] == 16
