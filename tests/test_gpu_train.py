"""GPU: fused training-loss kernel vs the oracle / the reference's golden values, and the trainer step."""
import numpy as np
import pytest
import torch

from tests.golden_utils import load
from tests.test_oracle_loss import _inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_fused_loss_matches_reference_golden_values_and_gradients():
    _need_gpu()
    from liuzhou_amd.train_loss import fused_policy_value_loss
    z = load("g11_loss.npz")
    for tag in ("a", "b"):
        alpha, anti, dw = (float(x) for x in z[f"{tag}_params"])
        l1, l2, l3, vl, mask, target, value, soft = _inputs(z)
        d = [x.detach().to(DEV).requires_grad_(True) for x in (l1, l2, l3, vl)]
        loss, parts = fused_policy_value_loss(*d, mask.to(DEV), target.to(DEV), value.to(DEV), soft.to(DEV),
                                              soft_label_alpha=alpha, anti_draw_penalty=anti, policy_draw_weight=dw)
        (loss * 3.0).backward()                       # an upstream factor, like the AMP loss scale
        np.testing.assert_allclose(loss.item(), z[f"{tag}_loss"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(parts["policy_loss"].item(), z[f"{tag}_policy_loss"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(parts["bucket_value_loss"].item(), z[f"{tag}_bucket_loss"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(parts["wdl_aux_loss"].item(), z[f"{tag}_wdl_aux"].mean(), rtol=1e-5, atol=1e-5)
        for g, key in zip(d, ("g1", "g2", "g3", "gv")):
            np.testing.assert_allclose(g.grad.cpu().numpy() / 3.0, z[f"{tag}_{key}"], rtol=1e-4, atol=1e-6)


def test_fused_loss_matches_oracle_on_a_large_random_batch_in_half_precision_inputs():
    _need_gpu()
    from liuzhou_amd.train_loss import fused_policy_value_loss
    from oracle.loss_oracle import policy_value_loss, combined_logits
    g = torch.Generator().manual_seed(5)
    B = 1537                                           # not a multiple of the 4 samples per workgroup
    raws = [torch.randn((B, 36), generator=g) for _ in range(3)]
    heads = [torch.log_softmax(r, 1) for r in raws]
    vl = (torch.randn((B, 101), generator=g) * 3).half().float()       # exactly representable in fp16
    mask = (torch.rand((B, 220), generator=g) < 0.15) & torch.isfinite(combined_logits(*heads))
    mask[:, 216] |= torch.rand(B, generator=g) < 0.1
    target = torch.rand((B, 220), generator=g) * mask
    target = target / target.sum(1, keepdim=True).clamp_min(1e-8)
    value = torch.randint(-1, 2, (B,), generator=g).float()
    soft = torch.rand(B, generator=g) * 2 - 1
    cpu = [h.clone().requires_grad_(True) for h in heads] + [vl.clone().requires_grad_(True)]
    want = policy_value_loss(*cpu, mask, target, value, soft, soft_label_alpha=0.25, anti_draw_penalty=0.0,
                             policy_draw_weight=0.5)
    want["loss"].backward()
    dev = [h.to(DEV).requires_grad_(True) for h in heads] + [vl.to(DEV).half().requires_grad_(True)]
    loss, parts = fused_policy_value_loss(*dev, mask.to(DEV), target.to(DEV), value.to(DEV), soft.to(DEV),
                                          soft_label_alpha=0.25, policy_draw_weight=0.5)
    loss.backward()
    np.testing.assert_allclose(loss.item(), want["loss"].item(), rtol=2e-5)
    assert dev[3].grad.dtype == torch.float16
    for a, b in zip(dev[:3], cpu[:3]):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dev[3].grad.float().cpu().numpy(), cpu[3].grad.numpy(), rtol=2e-3, atol=1e-7)


def test_trainer_step_learns_and_reports_the_reference_metrics(tmp_path):
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.train_bridge import train_network_from_tensors
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    torch.manual_seed(0)
    model = ChessNet(**MODEL_CONFIGS["b6c64"])
    stable_resnet_init(model, 20260314)
    model.to(DEV).eval()
    from liuzhou_amd.net_hip import FusedNet
    batch, stats = self_play_v1_gpu(FusedNet(model), num_games=64, mcts_simulations=16, temperature_init=1.0,
                                    temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0,
                                    device=DEV, max_game_plies=40, concurrent_games=64)
    assert batch.state_tensors.is_cuda                 # trained where it was produced: no CPU round trip
    opt_path = str(tmp_path / "adam.pt")
    model, m1 = train_network_from_tensors(model, batch, batch_size=256, epochs=2, lr=2e-3, device=DEV, use_amp=True,
                                           warmup_steps=4, soft_label_alpha=0.3, policy_draw_weight=0.5,
                                           optimizer_state_path=opt_path)
    assert [e["epoch"] for e in m1["epoch_stats"]] == [1, 2]
    e1, e2 = m1["epoch_stats"]
    assert e2["avg_loss"] < e1["avg_loss"]
    # batches whose scaled gradients overflow while the AMP scale settles are skipped and not counted, as in the reference
    assert e1["samples"] + 256 * e1["skipped_non_finite_grad_batches"] == batch.num_samples == 64 * 40
    assert e2["skipped_non_finite_grad_batches"] <= e1["skipped_non_finite_grad_batches"]
    for key in ("avg_policy_loss", "avg_value_loss", "avg_value_bucket_loss", "avg_wdl_aux_loss", "valid_policy_samples",
                "policy_weight_sum", "soft_alpha", "avg_soft_abs", "avg_mix_abs", "synced_batch_count",
                "skipped_non_finite_loss_batches", "skipped_non_finite_grad_batches", "filtered_non_finite_samples"):
        assert key in e1
    for key in ("num_samples", "num_samples_after_filter", "optimizer_loaded", "optimizer_lr_start", "optimizer_lr_final",
                "warmup_steps", "total_train_steps", "timing", "wdl_aux_loss_weight"):
        assert key in m1
    assert m1["optimizer_loaded"] is False and m1["warmup_steps"] == 4
    model, m2 = train_network_from_tensors(model, batch, batch_size=256, epochs=1, lr=1e-3, device=DEV,
                                           optimizer_state_path=opt_path)
    assert m2["optimizer_loaded"] is True and m2["epoch_stats"][0]["avg_loss"] < e1["avg_loss"]


def test_compact_trajectory_records_round_trip_exactly():
    """Wire format of the trajectory gather: 360-byte records reproduce the 5 tensors bit for bit."""
    _need_gpu()
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    from liuzhou_amd.trajectory_codec import RECORD_BYTES, pack_batch, unpack_records
    from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch
    torch.manual_seed(1)
    net = FusedNet(ChessNet(**MODEL_CONFIGS["b6c64"]).eval().to(DEV))
    batch, _ = self_play_v1_gpu(net, num_games=96, mcts_simulations=8, temperature_init=1.0, temperature_final=0.1,
                                temperature_threshold=10, exploration_weight=1.0, device=DEV, max_game_plies=150,
                                concurrent_games=96)
    assert batch.num_samples > 5000 and int(batch.state_tensors[:, 4:].sum(dim=(2, 3)).amax()) == 36
    phases = batch.state_tensors[:, 4:, 0, 0].argmax(1)
    assert len(torch.unique(phases)) >= 4                      # placement, movement, selections ... all occur
    rec = pack_batch(batch)
    assert rec.shape == (batch.num_samples, RECORD_BYTES) and rec.dtype == torch.uint8
    back = unpack_records(rec)
    for k in ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets"):
        a, b = getattr(batch, k), getattr(back, k)
        assert a.dtype == b.dtype and a.shape == b.shape
        assert torch.equal(a.view(torch.uint8) if a.dtype == torch.bool else a.view(torch.int32),
                           b.view(torch.uint8) if b.dtype == torch.bool else b.view(torch.int32)), k
    # rows that cannot be represented are refused loudly
    broken = TensorSelfPlayBatch(batch.state_tensors[:4] * 0.5, batch.legal_masks[:4], batch.policy_targets[:4],
                                 batch.value_targets[:4], batch.soft_value_targets[:4])
    with pytest.raises(RuntimeError, match="not representable"):
        pack_batch(broken)
    assert pack_batch(TensorSelfPlayBatch(batch.state_tensors[:0], batch.legal_masks[:0], batch.policy_targets[:0],
                                          batch.value_targets[:0], batch.soft_value_targets[:0])).shape == (0, RECORD_BYTES)


def test_streaming_trainer_from_shard_files(tmp_path):
    """Self-play stage output on disk -> resolve_shard_specs -> streaming DataLoader -> train_network_streaming."""
    _need_gpu()
    from liuzhou_amd import self_play_stage as S
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
    from liuzhou_amd.net_hip import FusedNet
    from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
    from liuzhou_amd.streaming import build_streaming_dataloader
    from liuzhou_amd.train_bridge import train_network_streaming
    torch.manual_seed(0)
    model = ChessNet(**MODEL_CONFIGS["b6c64"])
    stable_resnet_init(model, 20260314)
    model.to(DEV).eval()
    batch, _ = self_play_v1_gpu(FusedNet(model), num_games=48, mcts_simulations=8, temperature_init=1.0,
                                temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0, device=DEV,
                                max_game_plies=40, concurrent_games=48)
    path = str(tmp_path / "selfplay_iter_001.pt")
    S.save_sharded(path=path, samples=batch, stats_payload={}, metadata={}, num_shards=3)
    specs, total = S.resolve_shard_specs(path, [], 0)
    assert total == batch.num_samples == 48 * 40
    dl = build_streaming_dataloader(specs, batch_size=256, num_workers=0, pin_memory=False)
    model, m = train_network_streaming(model, dl, total_samples=total, batch_size=256, epochs=3, lr=2e-3, device=DEV,
                                       warmup_steps=2, soft_label_alpha=0.2)
    es = m["epoch_stats"]
    assert len(es) == 3 and es[-1]["avg_loss"] < es[0]["avg_loss"]
    assert m["streaming"] is True and m["est_batches_per_epoch"] == (total + 255) // 256
    # shards end on partial batches, so the loader yields more (smaller) batches than the estimate: no step is a no-op
    assert all(e["dataloader_exhausted_steps"] == 0 for e in es)
    assert es[0]["samples"] + 0 <= total
