"""Round 5: the full-length C3 product run -- ONE self_play_tree_gpu call with 16 384 games on 16 384 slots, 800 sims,
10x128, from the empty board to the end of the last game (the drain included) -- with and without the compact evaluation
lists.  Prints one JSON line per run; a progress line per ~60 s keeps the box from being taken for hung."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import clear_engine_cache, self_play_tree_gpu

def run(compact, games=16384, sims=800, model="b10c128", plies=512):
    """compact: True = lists in every search, False = dense launches only, None = the product default (dense while nearly
    all games are live, lists once the wave has drained by a network pass per CU)."""
    if compact is None:
        os.environ.pop("LZ_TREE_COMPACT", None)
    else:
        os.environ["LZ_TREE_COMPACT"] = "1" if compact else "0"
    dev = torch.device("cuda:0")
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS[model]).eval().to(dev))
    stop = False
    def tick():
        t0 = time.time()
        while not stop:
            time.sleep(30)
            print(f"[progress] compact={compact} {time.time() - t0:.0f} s", flush=True)
    th = threading.Thread(target=tick, daemon=True); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch, st = self_play_tree_gpu(net, num_games=games, mcts_simulations=sims, temperature_init=1.0, temperature_final=0.1,
                                   temperature_threshold=10, exploration_weight=1.0, device="cuda:0", add_dirichlet_noise=True,
                                   sample_moves=True, concurrent_games=games, max_game_plies=plies)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stop = True
    mc = st.mcts_counters
    print(json.dumps({"run": "self_play_tree_gpu full length", "compact_eval_lists": "auto" if compact is None else bool(compact), "games": games, "sims": sims,
                      "net": model, "positions": int(batch.num_samples), "elapsed_s": round(dt, 2),
                      "positions_per_s": round(batch.num_samples / dt, 1), "avg_game_length": round(st.avg_game_length, 2),
                      "plies_launched": mc.get("plies_launched"), "leaf_evals": mc.get("leaf_eval_count"),
                      "leaf_evals_per_s": round(mc.get("leaf_eval_count", 0) / dt), "loop_ms": mc.get("loop_ms"),
                      "host_wait_ms": mc.get("host_wait_ms"), "setup_ms": mc.get("setup_ms"),
                      "reuse_pruned": mc.get("reuse_pruned"), "reuse_dropped": mc.get("reuse_dropped"),
                      "edge_pool_refused": mc.get("edge_pool_refused")}), flush=True)
    del batch, net
    clear_engine_cache()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "compact"
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    sims = int(sys.argv[3]) if len(sys.argv) > 3 else 800
    if which == "auto":
        run(None, games, sims)
    if which in ("compact", "both"):
        run(True, games, sims)
    if which in ("dense", "both"):
        run(False, games, sims)
