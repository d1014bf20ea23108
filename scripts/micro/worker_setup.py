"""Where the worker's set-up time goes (checkpoint -> packed network on the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
from liuzhou_amd.net_pack import pack_model
from liuzhou_amd.self_play_worker import _infer_model
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
torch.manual_seed(0)
torch.save(ChessNet(**MODEL_CONFIGS["b10c128"]).state_dict(), "/tmp/lz_m.pt")
print("torch threads", torch.get_num_threads())
for rep in range(3):
    t = [time.perf_counter()]
    st = torch.load("/tmp/lz_m.pt", map_location="cpu"); t.append(time.perf_counter())
    m = _infer_model(st); t.append(time.perf_counter())
    m.load_state_dict(st, strict=True); t.append(time.perf_counter())
    m.to(dev).eval(); torch.cuda.synchronize(); t.append(time.perf_counter())
    p = pack_model(m); t.append(time.perf_counter())
    net = FusedNet(m, dev); torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["torch.load", "build module", "load_state_dict", "to(dev)", "pack_model alone", "FusedNet (incl. its own pack)"]
    print(rep, ", ".join(f"{n} {1e3*(b-a):.0f} ms" for n, a, b in zip(names, t, t[1:])), flush=True)
torch.set_num_threads(8)
t0 = time.perf_counter(); p = pack_model(m); print("pack_model with 8 threads", round(1e3 * (time.perf_counter() - t0)), "ms")
torch.set_num_threads(1)
t0 = time.perf_counter(); p = pack_model(m); print("pack_model with 1 thread", round(1e3 * (time.perf_counter() - t0)), "ms")
