"""Trainer-step measurements (f2): the fused loss kernel against its HBM roofline and against the PyTorch
composition it replaces, and whole training steps (forward + loss + backward + Adam, AMP) in samples/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
from liuzhou_amd.train_loss import fused_policy_value_loss
from liuzhou_amd.train_bridge import train_network_from_tensors
from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B = 16384
heads = [torch.log_softmax(torch.randn((B, 36), device=dev, generator=g), 1).requires_grad_(True) for _ in range(3)]
vl = torch.randn((B, 101), device=dev, generator=g).requires_grad_(True)
mask = torch.rand((B, 220), device=dev, generator=g) < 0.12
mask[:, 0] = True
target = torch.rand((B, 220), device=dev, generator=g) * mask
target = target / target.sum(1, keepdim=True)
value = torch.randint(-1, 2, (B,), device=dev, generator=g).float()
soft = torch.rand(B, device=dev, generator=g) * 2 - 1


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def fused():
    loss, _ = fused_policy_value_loss(*heads, vl, mask, target, value, soft, soft_label_alpha=0.3, policy_draw_weight=0.5)
    loss.backward()


def composed():                      # the same loss written as the ATen op chain the reference uses
    lp1, lp2, lpm = heads
    comb = torch.full((B, 220), float("-inf"), device=dev)
    comb[:, :36] = lp1
    idx = torch.arange(36, device=dev); r, c = idx // 6, idx % 6
    for d, (dr, dc) in enumerate(((-1, 0), (1, 0), (0, -1), (0, 1))):
        nr, nc = r + dr, c + dc
        ok = (nr >= 0) & (nr < 6) & (nc >= 0) & (nc < 6)
        dest = (nr * 6 + nc).clamp(0, 35)
        comb[:, 36 + idx * 4 + d] = torch.where(ok, lp2 + lp1[:, dest], torch.full_like(lp2, float("-inf")))
    comb[:, 180:216] = lpm
    comb[:, 216:] = 0
    masked = torch.where(mask, comb, torch.full_like(comb, float("-inf")))
    lse = torch.logsumexp(masked, 1, keepdim=True)
    logp = torch.where(mask, masked - lse, torch.zeros_like(comb)).clamp(min=-50.0)
    kl = -(target * logp).sum(1) + (target * target.clamp(min=1e-8).log()).sum(1)
    w = torch.where(value.abs() < 1e-8, 0.5, 1.0)
    pol = (kl * w).sum() / (w.sum() + 1e-8)
    mixed = (0.7 * value + 0.3 * soft).clamp(-1, 1)
    u = (mixed + 1.0) / 0.02
    lo = u.floor().long().clamp(0, 100); hi = (lo + 1).clamp(0, 100)
    frac = torch.where(hi == lo, torch.zeros_like(u), (u - lo.float()).clamp(0, 1))
    tgt = torch.zeros((B, 101), device=dev).scatter_add_(1, lo.view(-1, 1), (1 - frac).view(-1, 1)).scatter_add_(1, hi.view(-1, 1), frac.view(-1, 1))
    bucket = -(tgt * torch.log_softmax(vl, 1)).sum(1).mean()
    (pol + bucket).backward()


us_f, us_c = timed(fused), timed(composed)
bytes_per_sample = 3 * 36 * 4 + 101 * 4 + 220 + 220 * 4 + 8 + 3 * 36 * 4 + 101 * 4 + 16
print(f"loss fwd+bwd, B={B}: fused {us_f:.1f} us (whole autograd call; {B * bytes_per_sample / (us_f * 1e-6) / 1e9:.0f} GB/s of "
      f"{bytes_per_sample} B/sample algorithmic), ATen composition {us_c:.1f} us -> {us_c / us_f:.1f}x", flush=True)

for name in ("b6c64", "b10c128"):
    torch.manual_seed(0)
    model = ChessNet(**MODEL_CONFIGS[name]); stable_resnet_init(model, 20260314); model.to(dev)
    n = 65536
    batch = TensorSelfPlayBatch(state_tensors=(torch.rand((n, 11, 6, 6), device=dev) < 0.2).float(),
                                legal_masks=mask.repeat(n // B, 1), policy_targets=target.repeat(n // B, 1),
                                value_targets=value.repeat(n // B), soft_value_targets=soft.repeat(n // B))
    train_network_from_tensors(model, batch, batch_size=4096, epochs=1, device="cuda:0")      # MIOpen warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, m = train_network_from_tensors(model, batch, batch_size=4096, epochs=2, device="cuda:0")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"train step {name}: {2 * n / dt:,.0f} samples/s (batch 4096, AMP fp16, Adam, {m['total_train_steps']} steps, "
          f"loss {m['epoch_stats'][0]['avg_loss']:.3f} -> {m['epoch_stats'][1]['avg_loss']:.3f})", flush=True)
