"""Host time per `v0_core` operator call through the two bindings of the C ABI (compiled PyBind11 layer vs ctypes layer):
tiny batches, so that the time is the binding's (argument checks, output allocation, the launch), not the kernel's."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from liuzhou_amd import v0_core
dev = sys.argv[1] if len(sys.argv) > 1 else ("cuda:0" if torch.cuda.is_available() else "cpu")
B = 8
st = [torch.zeros(B, 6, 6, dtype=torch.int8, device=dev), torch.zeros(B, 6, 6, dtype=torch.bool, device=dev),
      torch.zeros(B, 6, 6, dtype=torch.bool, device=dev)] + [torch.ones(B, dtype=torch.int64, device=dev) for _ in range(9)]
codes = torch.tensor([[1, i, -1, -1] for i in range(B)], dtype=torch.int32, device=dev)
parents = torch.arange(B, device=dev)
lp = torch.log_softmax(torch.randn(B, 36, device=dev), 1)
out = {"device": dev, "batch": B}
for kind in ("native", "python"):
    try:
        ns = v0_core.binding(kind)
    except RuntimeError as exc:
        out[kind] = str(exc); continue
    mask, meta = ns.encode_actions_fast(*st[:10], 36, 144, 36, 4)
    ops = {"states_to_model_input": lambda: ns.states_to_model_input(*st[:5]),
           "encode_actions_fast": lambda: ns.encode_actions_fast(*st[:10], 36, 144, 36, 4),
           "batch_apply_moves": lambda: ns.batch_apply_moves(*st, codes, parents),
           "project_policy_logits_fast": lambda: ns.project_policy_logits_fast(lp, lp, lp, mask, 36, 144, 36, 4)}
    res = {}
    for name, fn in ops.items():
        for _ in range(200): fn()
        if dev != "cpu": torch.cuda.synchronize()
        n = 3000
        t0 = time.perf_counter()
        for _ in range(n): fn()
        dt = time.perf_counter() - t0                       # host time to ISSUE n calls (the device runs behind)
        if dev != "cpu": torch.cuda.synchronize()
        res[name] = round(dt / n * 1e6, 2)
    out[kind] = res
print(json.dumps(out))
