"""Round-5 experiment: where the worker's wall time goes (streamed vs the reference-style chunk loop, tree / root)."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.self_play_worker import run_self_play_worker
from liuzhou_amd.tree_engine import clear_engine_cache

def one(backend, model_name, games, slots, sims, stream, extra_env=None, opening=0):
    os.environ["LZ_WORKER_STREAM"] = "1" if stream else "0"
    for k, v in (extra_env or {}).items():
        os.environ[k] = v
    tmp = tempfile.mkdtemp(prefix="lz_exp_worker_")
    try:
        torch.manual_seed(1)
        torch.save(ChessNet(**MODEL_CONFIGS[model_name]).state_dict(), os.path.join(tmp, "m.pt"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run_self_play_worker(worker_idx=0, shard_device="cuda:0", shard_games=games, seed=9973,
                                   model_state_path=os.path.join(tmp, "m.pt"), output_path=os.path.join(tmp, "w.pt"),
                                   mcts_simulations=sims, temperature_init=1.0, temperature_final=0.1, temperature_threshold=10,
                                   exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, soft_value_k=2.0,
                                   opening_random_moves=opening, max_game_plies=512, concurrent_games_per_device=slots,
                                   chunk_output_dir=tmp, chunk_file_prefix="w", search_backend=backend)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        man = torch.load(os.path.join(tmp, "w.pt"), map_location="cpu")
        st = man["stats"]
        print(json.dumps({"backend": backend, "stream": stream, "env": extra_env, "elapsed_s": round(dt, 2),
                          "positions_per_s": round(res["num_samples"] / dt), "positions": res["num_samples"], "opening": opening,
                          "runner_elapsed": round(st.get("elapsed_sec", 0), 2), "chunks": res["saved_chunks"],
                          "timing_ms": {k: round(v) for k, v in st.get("step_timing_ms", {}).items()},
                          "counters": st.get("mcts_counters")}), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        clear_engine_cache()
        for k in (extra_env or {}):
            os.environ.pop(k, None)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "tree"
    if which == "reference_shape":
        # one worker of scripts/big_train_v1.sh's defaults: 522 488 games / 8 devices ~ 65 536 games on 8 192 slots,
        # opening_random_moves 6, 10x128 (sims 1 024 here)
        one("cuda_root", "b10c128", 65536, 8192, 1024, True, opening=6)
    if which == "opening":
        for op in (6, 0, 6, 0, 6, 0):
            one("portable", "b6c64", 8192, 4096, 200, True, opening=op)
    if which == "tree":
        pass
        one("cuda_root", "b10c128", 32768, 16384, 1024, True)
        one("cuda_root", "b10c128", 32768, 16384, 1024, True)
        one("cuda_root", "b10c128", 32768, 16384, 1024, False)
    else:
        one("cuda_root", "b10c128", 32768, 16384, 1024, True)
        one("cuda_root", "b10c128", 32768, 16384, 1024, True, {"LZ_WORKER_SEGMENT_GAMES": "2048"})
