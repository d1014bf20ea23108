"""Round 6: why did bench.py's `self_play_tree_gpu@C2` leg run at 31.9 ms per ply (halves serialised) in one default
invocation?  The runner leg alone, one child process per (stream pair mode, split step) setting, plus -- in-process -- after
a steady-state C2 population has been built and dropped (what bench.py's earlier legs leave behind)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(after_harness: bool):
    import torch
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.tree_engine import clear_engine_cache, self_play_tree_gpu, SteadyStateTreeSelfPlay
    dev = torch.device("cuda:0")
    torch.manual_seed(20260314)
    if after_harness:
        for prec in ("fp16", "fp16x3"):
            pop = SteadyStateTreeSelfPlay(FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev), precision=prec), 4096, sims=200,
                                          device=dev, seed=9973, reuse_tree=True, dual_stream=True, arena_rows=4096 * 40)
            pop.preroll(120); pop.prepare()
            for _ in range(5):
                pop.step()
            torch.cuda.synchronize()
            del pop
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    out = []
    for call in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        batch, st = self_play_tree_gpu(net, num_games=8192, mcts_simulations=200, temperature_init=1.0, temperature_final=0.1,
                                       temperature_threshold=10, exploration_weight=1.0, device="cuda:0", add_dirichlet_noise=True,
                                       sample_moves=True, opening_random_moves=6, concurrent_games=4096, max_game_plies=512)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        mc = st.mcts_counters
        out.append({"positions_per_s": round(batch.num_samples / dt), "ms_per_ply": round(dt * 1e3 / max(1, mc.get("plies_launched", 1)), 2),
                    "redraws": mc.get("stream_redraws")})
        del batch
    print(json.dumps({"pair": os.environ.get("LZ_STREAM_PAIR"), "split": os.environ.get("LZ_TREE_SPLIT"),
                      "after_harness_legs": after_harness, "calls": out}), flush=True)
    clear_engine_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2] == "1")
    else:
        for after in ("0", "1"):
            for pair in ("priority", "probe"):
                for split in ("1", "0"):
                    env = dict(os.environ, LZ_STREAM_PAIR=pair, LZ_TREE_SPLIT=split)
                    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", after], env=env)
