"""What fp16 network arithmetic does to the search's bit-exact claim (VERDICT r02, weak #1a).

(i)  The fused fp16-MFMA kernel against `torch.autocast(float16)` of the same module on PyTorch-ROCm -- the reference's
     actual inference mode (v1/python/mcts_gpu.py:640-646) -- and both against the fp32 module, on real positions (g1):
     max |delta log-prob| on the legal-relevant range, max |delta prob|, max |delta value|.
(ii) The same positions searched twice with the production engine (captured search, same injected Dirichlet noise,
     deterministic picks): network in fused fp16 vs the fp32-operand kernel (`LzNetDesc.flags` bit 2).  Reported:
     fraction of roots with identical visit counts, identical most-visited move, max / mean L1 distance of the visit
     policies.  The tree arithmetic is the same double-precision code in both runs; only the evaluations differ.

    python scripts/exp/exp_fp16_effect.py [b6c64|b10c128] [games] [sims]      -> one JSON line
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.tree_engine import PortableTreeMCTS
from tests.golden_utils import FIELDS, load, states
from tests.tree_parity import engine_visits, to_gpu_batch

DEV = torch.device("cuda:0")


def positions(n, seed):
    st = states(load("g1_rules.npz"), "s")
    idx = np.random.default_rng(seed).integers(0, st["board"].shape[0], n)
    return {f: np.ascontiguousarray(np.asarray(st[f])[idx]) for f in FIELDS}


def net_deltas(model, batch):
    from liuzhou_amd.mcts_gpu import states_to_model_input
    x = states_to_model_input(batch)
    with torch.inference_mode():
        r32 = model(x)
        v32 = bucket_logits_to_scalar(r32[3])
        with torch.autocast("cuda", dtype=torch.float16):
            r16 = model(x)
        r16 = tuple(t.float() for t in r16)
        v16 = bucket_logits_to_scalar(r16[3])
    fused = FusedNet(model)
    f = fused(x)
    vf = fused.last_value
    out = {}
    for name, a, va in (("fused_vs_autocast", r16, v16), ("fused_vs_fp32", r32, v32)):
        dl = max(float((f[k] - a[k]).abs()[a[k] > -12.0].max().item()) for k in range(3))     # log-probs that matter
        dp = max(float((f[k].exp() - a[k].exp()).abs().max().item()) for k in range(3))
        out[name] = {"max_abs_dlogp": round(dl, 6), "max_abs_dprob": round(dp, 6),
                     "max_abs_dvalue": round(float((vf - va).abs().max().item()), 6)}
    dl = max(float((r16[k] - r32[k]).abs()[r32[k] > -12.0].max().item()) for k in range(3))
    dp = max(float((r16[k].exp() - r32[k].exp()).abs().max().item()) for k in range(3))
    out["autocast_vs_fp32"] = {"max_abs_dlogp": round(dl, 6), "max_abs_dprob": round(dp, 6),
                               "max_abs_dvalue": round(float((v16 - v32).abs().max().item()), 6)}
    return out


def search_visits(model, batch, sims, precision, noise):
    net = FusedNet(model, precision=precision)
    B = int(batch.board.shape[0])
    m = PortableTreeMCTS(net, B, sims, DEV, add_dirichlet_noise=True, sample_moves=False, use_graph=True)
    m.injected_noise = noise
    m.search_batch(batch, temperatures=torch.ones((B,), device=DEV))
    torch.cuda.synchronize(DEV)
    return engine_visits(m.engine)[0]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "b6c64"
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    sims = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    torch.manual_seed(20260314)
    model = ChessNet(**MODEL_CONFIGS[name]).eval().to(DEV)
    res = {"net": name, "games": games, "sims": sims}
    res["network"] = net_deltas(model, to_gpu_batch(positions(2048, 11), DEV))
    st = positions(games, 12)
    batch = to_gpu_batch(st, DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    noise = torch._standard_gamma(torch.full((games, 80), 0.3, device=DEV), generator=g)
    v16 = search_visits(model, batch, sims, "fp16", noise)
    v32 = search_visits(model, batch, sims, "fp32", noise)
    v16b = search_visits(model, batch, sims, "fp16", noise)
    live = v32.sum(1) > 0
    same = (v16 == v32).all(1)[live]
    arg = (v16.argmax(1) == v32.argmax(1))[live]
    p16 = v16[live] / np.maximum(v16[live].sum(1, keepdims=True), 1)
    p32 = v32[live] / np.maximum(v32[live].sum(1, keepdims=True), 1)
    l1 = np.abs(p16 - p32).sum(1)
    res["search"] = {"roots": int(live.sum()), "identical_visit_counts": round(float(same.mean()), 4),
                     "identical_most_visited": round(float(arg.mean()), 4), "policy_l1_max": round(float(l1.max()), 4),
                     "policy_l1_mean": round(float(l1.mean()), 5),
                     "fp16_run_to_run_identical": bool((v16 == v16b).all())}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
