"""Trainer step in the staged loop: what a NEW last-batch size costs (every iteration's sample count is different, so the
remainder batch has a shape MIOpen has not seen) with PyTorch's default MIOpen find path and with
`torch.backends.miopen.immediate = True`, and the steady-state samples/s either way.

    python scripts/exp/exp_train_shapes.py [b6c64|b10c128]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, stable_resnet_init
from liuzhou_amd.train_bridge import train_network_from_tensors
from liuzhou_amd.trajectory_buffer import TensorSelfPlayBatch

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "b10c128"
g = torch.Generator(device=dev).manual_seed(1)


def batch(n):
    mask = torch.rand((n, 220), device=dev, generator=g) < 0.12
    mask[:, 0] = True
    target = torch.rand((n, 220), device=dev, generator=g) * mask
    return TensorSelfPlayBatch(state_tensors=(torch.rand((n, 11, 6, 6), device=dev, generator=g) < 0.2).float(), legal_masks=mask,
                               policy_targets=target / target.sum(1, keepdim=True),
                               value_targets=torch.randint(-1, 2, (n,), device=dev, generator=g).float(),
                               soft_value_targets=torch.rand(n, device=dev, generator=g) * 2 - 1)


out = {"net": name}
for immediate in (False, True):
    torch.backends.miopen.immediate = immediate
    torch.manual_seed(0)
    model = ChessNet(**MODEL_CONFIGS[name]); stable_resnet_init(model, 20260314); model.to(dev)
    rows = []
    for n in (65536, 65536, 65536 + 1234, 65536 + 3001, 65536 + 777, 131072):
        b = batch(n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        train_network_from_tensors(model, b, batch_size=4096, epochs=1, device="cuda:0")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rows.append({"samples": n, "sec": round(dt, 3), "samples_per_s": round(n / dt, 0)})
    out["immediate" if immediate else "default_find"] = rows
print(json.dumps(out), flush=True)
