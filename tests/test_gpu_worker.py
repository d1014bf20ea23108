"""GPU: run_self_play_worker contract (tests/v1/test_v1_tensor_pipeline_smoke.py:132-176 in the reference):
chunk payloads + worker manifest, for both search backends."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend,chunk_target_bytes", [("cuda_root", 0), ("cuda_root", 1 << 16), ("portable", 0)])
def test_worker_emits_chunk_manifest(tmp_path, backend, chunk_target_bytes):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.self_play_worker import run_self_play_worker
    torch.manual_seed(3)
    model = ChessNet(**MODEL_CONFIGS["b6c64"])
    state_path = tmp_path / "model_state.pt"
    torch.save(model.state_dict(), state_path)
    manifest_path = tmp_path / f"worker_manifest_{backend}_{chunk_target_bytes}.pt"
    row = run_self_play_worker(
        worker_idx=0, shard_device="cuda:0", shard_games=3, seed=7, model_state_path=str(state_path),
        output_path=str(manifest_path), mcts_simulations=4, temperature_init=1.0, temperature_final=0.1,
        temperature_threshold=4, exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, soft_value_k=2.0,
        opening_random_moves=2, max_game_plies=64, concurrent_games_per_device=2, soft_label_alpha=0.0,
        target_samples_per_shard=0, chunk_target_bytes=int(chunk_target_bytes), chunk_output_dir=str(tmp_path),
        chunk_file_prefix=f"worker_{backend}_{chunk_target_bytes}", search_backend=backend)
    payload = torch.load(manifest_path, map_location="cpu")
    assert row["output_path"] == str(manifest_path) and row["games"] == 3
    assert set(row) == {"worker_idx", "device", "games", "output_path", "num_samples", "saved_chunks"}
    assert payload["payload_format"] == "v1_worker_chunk_manifest"
    assert int(payload["num_shards"]) >= 1 and len(payload["shard_files"]) == int(payload["num_shards"])
    assert sum(payload["shard_sizes"]) == payload["num_samples"] == row["num_samples"] > 0
    for key in ("stats", "value_target_summary", "soft_value_target_summary", "mixed_value_target_summary", "metadata",
                "avg_bytes_per_sample", "chunk_target_bytes", "version"):
        assert key in payload
    assert payload["avg_bytes_per_sample"] == 2692
    total = 0
    for name in payload["shard_files"]:
        shard = torch.load(tmp_path / str(name), map_location="cpu")
        assert set(shard) == {"state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets",
                              "stats", "metadata"}
        assert shard["metadata"]["payload_format"] == "v1_sharded_shard"
        n = shard["state_tensors"].shape[0]
        total += n
        assert shard["state_tensors"].shape[1:] == (11, 6, 6) and shard["legal_masks"].shape == (n, 220)
        assert torch.isfinite(shard["value_targets"]).all()
        pol = shard["policy_targets"]
        assert torch.allclose(pol.sum(1), torch.ones(n), atol=1e-4)
    assert total == payload["num_samples"]
    assert payload["stats"]["num_games"] == 3.0


@pytest.mark.parametrize("arch,want", [
    (dict(trunk_channels=64, num_blocks=2, policy_channels=32, value_channels=32), "torch"),
    (dict(trunk_channels=64, num_blocks=20), "fused_f16"),
    (dict(trunk_channels=32, num_blocks=2), "torch"),
    (dict(trunk_channels=128, num_blocks=1, value_mlp_channels=64), "torch")])
@pytest.mark.parametrize("backend", ["cuda_root", "portable"])
def test_worker_names_its_evaluator_and_never_dies_on_a_shape(tmp_path, backend, arch, want):
    """A checkpoint the trainer can produce must play: shapes the fused kernel is not built for (other head sizes, other
    widths -- src/neural_network.py:213-246 is generic in all of them) are evaluated by the module itself, deeper
    64 / 128-channel nets by the kernel, and the manifest says which evaluator ran (`metadata.evaluator`)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet
    from liuzhou_amd.self_play_worker import run_self_play_worker
    torch.manual_seed(5)
    state_path = tmp_path / "model_state.pt"
    torch.save(ChessNet(**arch).state_dict(), state_path)
    manifest_path = tmp_path / "worker_manifest.pt"
    row = run_self_play_worker(
        worker_idx=0, shard_device="cuda:0", shard_games=3, seed=7, model_state_path=str(state_path),
        output_path=str(manifest_path), mcts_simulations=4, temperature_init=1.0, temperature_final=0.1,
        temperature_threshold=4, exploration_weight=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, soft_value_k=2.0,
        opening_random_moves=2, max_game_plies=20, concurrent_games_per_device=3, soft_label_alpha=0.0,
        chunk_output_dir=str(tmp_path), chunk_file_prefix="w", search_backend=backend)
    payload = torch.load(manifest_path, map_location="cpu")
    meta = payload["metadata"]
    assert meta["evaluator"] == want and (("evaluator_reason" in meta) == (want == "torch"))
    assert row["num_samples"] == payload["num_samples"] > 0
    for name in payload["shard_files"]:
        shard = torch.load(tmp_path / str(name), map_location="cpu")
        assert shard["metadata"]["evaluator"] == want
        pol = shard["policy_targets"]
        assert torch.allclose(pol.sum(1), torch.ones(pol.shape[0]), atol=1e-4)
        assert torch.isfinite(shard["value_targets"]).all()


@pytest.mark.parametrize("backend", ["cuda_root", "portable"])
def test_selfplay_stage_cli_two_worker_processes(tmp_path, backend):
    """scripts/selfplay_stage.py end to end: two spawned worker processes (both on cuda:0 here), chunk files, the
    sharded manifest and the stats json; the loader reads every sample back."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import os
    import subprocess
    import sys
    from liuzhou_amd.self_play_stage import load_self_play_payload, resolve_shard_specs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "selfplay_iter_001.pt"
    stats_json = tmp_path / "stats.json"
    cmd = [sys.executable, os.path.join(root, "scripts", "selfplay_stage.py"), "--pipeline", "v1", "--stage", "selfplay",
           "--devices", "cuda:0,cuda:0", "--self_play_games", "10", "--mcts_simulations", "6", "--model", "b6c64",
           "--self_play_concurrent_games", "4", "--max_game_plies", "24", "--search_backend", backend,
           "--self_play_output", str(out), "--self_play_stats_json", str(stats_json),
           "--self_play_iteration_seed", "2", "--train_devices", "cuda:0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    manifest = torch.load(out, map_location="cpu", weights_only=False)
    assert manifest["payload_format"] == "v1_sharded_manifest" and manifest["num_shards"] >= 2
    assert {f.split(".")[1] for f in manifest["shard_files"]} == {"w00", "w01"}
    batch, stats, meta = load_self_play_payload(str(out))
    assert batch.num_samples == manifest["num_samples"] == 10 * 24          # every game hits the ply cap
    assert int(stats["num_games"]) == 10 and meta["search_backend"] == backend
    assert torch.allclose(batch.policy_targets.sum(1), torch.ones(batch.num_samples), atol=1e-4)
    assert bool((batch.policy_targets[~batch.legal_masks] == 0).all())
    specs, total = resolve_shard_specs(str(out), [], 0)
    assert total == batch.num_samples
    js = json.load(open(stats_json))
    assert js["num_samples"] == batch.num_samples


def test_staged_loop_single_process(tmp_path):
    """C5 staged loop (self-play -> gather -> train -> checkpoint hand-off), one process = one GPU doing both roles."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "staged_loop.py"), "--iterations", "2",
                          "--games-per-gpu", "64", "--sims", "8", "--max-game-plies", "40", "--batch-size", "512"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert len(out["iterations"]) == 2 and out["iterations"][0]["positions"] == 64 * 40
    assert out["iterations"][1]["train_samples"] > 0 and out["steady_state_positions_per_sec"] > 0
    assert out["iterations"][1]["avg_loss"] is not None


def test_staged_loop_searches_with_weights_of_three_real_training_iterations():
    """Four iterations of the staged loop with a learning rate that moves the weights: iterations 2..4 search with
    checkpoints that went through 1..3 real training passes and `FusedNet.refresh`; after every hand-off the refreshed
    fp16 kernel is compared with the fp32 module and with torch.autocast(float16) on that iteration's positions."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "staged_loop.py"), "--iterations", "4",
                          "--games-per-gpu", "128", "--sims", "16", "--max-game-plies", "48", "--batch-size", "512",
                          "--epochs", "2", "--lr", "0.004", "--net-check", "2048"],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    its = out["iterations"]
    assert len(its) == 4 and len({it["weights_digest"] for it in its}) == 4          # every hand-off carried new weights
    for it in its:
        nc = it["net_check"]
        assert nc["finite"] and nc["positions"] == 2048
        assert nc["max_dprob_fused_vs_fp32"] <= 4.0 * nc["max_dprob_autocast_vs_fp32"] + 1e-5, nc
        assert nc["max_dvalue_fused_vs_fp32"] <= 4.0 * nc["max_dvalue_autocast_vs_fp32"] + 1e-5, nc
        assert nc["argmax_agreement_min_head"] >= 0.99, nc
        assert it["positions"] > 0 and it["avg_loss"] is not None
    assert its[-1]["avg_loss"] < its[0]["avg_loss"]
    print("net_check per iteration:", [(round(it["net_check"]["max_dprob_fused_vs_fp32"], 7),
                                        round(it["net_check"]["max_dprob_autocast_vs_fp32"], 7),
                                        round(it["net_check"]["max_abs_logprob_fp32"], 2)) for it in its])


def test_one_full_iteration_selfplay_train_eval_through_the_clis(tmp_path):
    """selfplay_stage.py -> train_stage.py (streaming) -> eval_arena.py: the three stages of one big_train_v1 iteration."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = lambda *cmd: subprocess.run([sys.executable, *cmd], capture_output=True, text=True, timeout=600)
    sp = tmp_path / "selfplay_iter_001.pt"
    r = run(os.path.join(root, "scripts", "selfplay_stage.py"), "--devices", "cuda:0", "--self_play_games", "48",
            "--mcts_simulations", "8", "--model", "b6c64", "--self_play_concurrent_games", "48", "--max_game_plies", "40",
            "--self_play_output", str(sp), "--self_play_chunk_target_bytes", "2000000")
    assert r.returncode == 0, r.stdout + r.stderr
    ck_dir = tmp_path / "ck"
    metrics = tmp_path / "train.json"
    r = run(os.path.join(root, "scripts", "train_stage.py"), "--stage", "train", "--self_play_input", str(sp),
            "--streaming_load", "1", "--streaming_workers", "2", "--batch_size", "256", "--epochs", "2", "--lr", "0.002",
            "--model", "b6c64", "--checkpoint_dir", str(ck_dir), "--checkpoint_name", "model_iter_001.pt",
            "--metrics_output", str(metrics), "--optimizer_state_path", str(tmp_path / "adam.pt"), "--train_devices", "cuda:0")
    assert r.returncode == 0, r.stdout + r.stderr
    m = json.load(open(metrics))[0]
    assert m["streaming"] is True and m["train_avg_loss"] is not None and os.path.exists(m["checkpoint"])
    es = m["train_bridge"]["epoch_stats"]
    assert es[-1]["avg_loss"] < es[0]["avg_loss"] and es[0]["samples"] > 0
    ck = torch.load(ck_dir / "model_iter_001.pt", map_location="cpu", weights_only=False)
    assert "model_state_dict" in ck and os.path.exists(tmp_path / "adam.pt")
    ev = tmp_path / "eval.json"
    r = run(os.path.join(root, "scripts", "eval_arena.py"), "--challenger_checkpoint", str(ck_dir / "model_iter_001.pt"),
            "--eval_games_vs_random", "16", "--mcts_simulations", "8", "--output_json", str(ev), "--seed", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    e = json.load(open(ev))
    assert e["vs_random"]["total_games"] == 16
