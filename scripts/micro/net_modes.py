"""The three arithmetic modes of the fused network kernel side by side: evaluations/s (sustained ~2 s per shape) and the
largest deviation from the fp32 PyTorch module over every output (log-probs, 101 value logits, scalar value).
fp16   = production (fp16 MFMA operands, fp32 accumulate: the reference's autocast inference mode)
fp16x3 = split operands (two fp16 numbers per operand = 22 bits, three fp16 MFMAs per product; round 6)
fp32   = fp32 operands (v_mfma_f32_16x16x4_f32, 1/16 of the fp16 MFMA rate)
usage: python scripts/micro/net_modes.py            (one JSON line per net and mode)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
from liuzhou_amd.net_hip import FusedNet

dev = torch.device("cuda:0")
for name, N in (("b6c64", 4096), ("b10c128", 4096)):
    torch.manual_seed(20260314)
    m = ChessNet(**MODEL_CONFIGS[name]).eval().to(dev)
    g = torch.Generator().manual_seed(5)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    x = (torch.rand(N, 11, 6, 6, generator=g) < 0.3).float().to(dev)
    with torch.inference_mode():
        r1, r2, rm, rv = m(x)
        rval = bucket_logits_to_scalar(rv)
    for mode in ("fp16", "fp16x3", "fp32"):
        f = FusedNet(m, precision=mode)
        lp1, lp2, lpm, vl = f(x)
        err = {"log_probs": max((a - b).abs().max().item() for a, b in ((lp1, r1), (lp2, r2), (lpm, rm))),
               "probs": max((a.exp() - b.exp()).abs().max().item() for a, b in ((lp1, r1), (lp2, r2), (lpm, rm))),
               "value_logits": (vl - rv).abs().max().item(), "value": (f.last_value - rval).abs().max().item()}
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(5):
                f(x, want_logits=False)
            torch.cuda.synchronize(); n += 5
        dt = time.perf_counter() - t0
        print(json.dumps({"net": name, "mode": mode, "evals_per_launch": N, "us_per_launch": round(dt / n * 1e6, 1),
                          "evals_per_s": round(n * N / dt), "tflops_algorithmic": round(n * N / dt * f.flops_per_eval / 1e12, 1),
                          "max_abs_err_vs_fp32_module": {k: float(f"{v:.3g}") for k, v in err.items()}}), flush=True)
