"""Round 6: what does the overlap watch see inside the streamed C2 worker (it re-draws its pair once per run)?"""
import os, sys, tempfile, shutil, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LZ_WATCH_DEBUG"] = "1"
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.self_play_worker import run_self_play_worker
tmp = tempfile.mkdtemp(prefix="lz_ww_")
torch.manual_seed(20260314)
torch.save(ChessNet(**MODEL_CONFIGS["b6c64"]).state_dict(), os.path.join(tmp, "m.pt"))
t0 = time.perf_counter()
res = run_self_play_worker(worker_idx=0, shard_device="cuda:0", shard_games=8192, seed=9973, model_state_path=os.path.join(tmp, "m.pt"),
                           output_path=os.path.join(tmp, "w.pt"), mcts_simulations=200, temperature_init=1.0, temperature_final=0.1,
                           temperature_threshold=10, exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, soft_value_k=2.0,
                           opening_random_moves=6, max_game_plies=512, concurrent_games_per_device=4096, chunk_output_dir=tmp,
                           chunk_file_prefix="w", search_backend="portable")
dt = time.perf_counter() - t0
man = torch.load(os.path.join(tmp, "w.pt"), map_location="cpu")
print("positions/s", round(res["num_samples"] / dt), {k: v for k, v in man["stats"]["mcts_counters"].items() if "stream" in k}, file=sys.stderr)
shutil.rmtree(tmp, ignore_errors=True)
