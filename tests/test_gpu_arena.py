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


def test_arena_reproduces_the_reference_arena_worker():
    """g14: the reference's own arena worker (`scripts/eval_checkpoint.py::_eval_worker_v1`, backend v1, two tiny
    checkpoints, deterministic picks, no random openings) recorded on CPU -- outcome tuple, every game's move sequence,
    and both agents' network evaluations.  The device arena, with the two networks replaced by tables of those recorded
    evaluations (host-independent), must play the same moves and count the same results."""
    _need_gpu()
    import numpy as np
    from liuzhou_amd.eval_arena import RootSearchAgent, play_matches
    from tests.golden_utils import load
    z = load("g14_eval_arena.npz")
    G, sims = (int(x) for x in z["config"])

    def fnv64(rows):
        h = np.full(rows.shape[0], 0xCBF29CE484222325, np.uint64)
        for j in range(rows.shape[1]):
            h = (h ^ rows[:, j].astype(np.uint64)) * np.uint64(0x100000001B3)
        return h

    class TableNet(torch.nn.Module):
        """forward(planes) -> (log_p1, log_p2, log_pmc, value[N,1]) looked up by the packed planes."""
        def __init__(self, tag):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1))
            self.val = {int(k): float(v) for k, v in zip(z[f"{tag}_keys"], z[f"{tag}_values"])}
            self.heads = {int(k): h for k, h in zip(z[f"{tag}_root_keys"], z[f"{tag}_root_heads"])}
            self.misses = 0

        def forward(self, x):
            keys = fnv64(np.packbits(x.detach().float().cpu().numpy().astype(bool).reshape(x.shape[0], -1), axis=1))
            heads = np.zeros((x.shape[0], 108), np.float32); val = np.zeros((x.shape[0], 1), np.float32)
            for i, k in enumerate(keys.tolist()):
                if k in self.val:
                    val[i, 0] = self.val[k]
                else:
                    self.misses += 1
                if k in self.heads:
                    heads[i] = self.heads[k]
            t = lambda a: torch.from_numpy(a).to(x.device)
            return t(heads[:, 0:36].copy()), t(heads[:, 36:72].copy()), t(heads[:, 72:108].copy()), t(val)

    nets = {tag: TableNet(tag) for tag in ("chall", "opp")}
    agents = {tag: RootSearchAgent(nets[tag], DEV, sims, temperature=0.1, sample_moves=False) for tag in nets}
    stats = play_matches(agents["chall"], agents["opp"], G, DEV, record_moves=True)
    want = [int(v) for v in z["result"]]
    cb = stats.color_breakdown
    got = [stats.wins, stats.losses, stats.draws, cb["black"]["wins"], cb["black"]["losses"], cb["black"]["draws"],
           cb["white"]["wins"], cb["white"]["losses"], cb["white"]["draws"]]
    moves = stats.move_log.cpu().numpy()
    ref = z["moves"]
    L = ref.shape[1]
    assert moves.shape[1] >= L
    assert np.array_equal(moves[:, :L], ref), "the games diverge from the reference arena's move sequences"
    assert (moves[:, L:] == -1).all()
    assert got == want, (got, want)
    assert nets["chall"].misses == 0 and nets["opp"].misses == 0, "a position the reference never evaluated was searched"
