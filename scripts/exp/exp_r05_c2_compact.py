"""Round 5: does the compact evaluation list pay at C2 (one network pass per CU per half)?  Product runner, 16 384 games on
4 096 slots, 200 sims, 6x64, with LZ_TREE_COMPACT = 0 / 1."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import clear_engine_cache, self_play_tree_gpu
dev = torch.device("cuda:0")
for compact in ("0", "1", "0", "1"):
    os.environ["LZ_TREE_COMPACT"] = compact
    torch.manual_seed(20260314)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(dev))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch, st = self_play_tree_gpu(net, num_games=16384, mcts_simulations=200, temperature_init=1.0, temperature_final=0.1,
                                   temperature_threshold=10, exploration_weight=1.0, device="cuda:0", add_dirichlet_noise=True,
                                   sample_moves=True, concurrent_games=4096, max_game_plies=512)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mc = st.mcts_counters
    print(json.dumps({"compact": compact, "positions": int(batch.num_samples), "elapsed_s": round(dt, 2),
                      "positions_per_s": round(batch.num_samples / dt), "plies": mc.get("plies_launched"),
                      "leaf_evals": mc.get("leaf_eval_count"), "lists": mc.get("compact_eval_lists")}), flush=True)
    del batch, net
    clear_engine_cache()
