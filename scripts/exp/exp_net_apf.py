#!/usr/bin/env python3
"""Round-4 experiment: weight fragments of the network kernel fetched one K step ahead (-DLZ_NET_APF=1) vs two
(-DLZ_NET_APF=2).  Sustained evaluations/s of the stand-alone kernel at the bench's launch shapes, outputs compared bit for
bit, one child process per build of the library (LZ_HIP_LIB).  Build the other variant by hand:
  cd liuzhou_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
      -DLZ_NET_APF=<1|2> -o ../_exp/liblz_APF<1|2>.so lz_ops.hip lz_engine.hip lz_net.hip lz_net_f32.hip lz_train.hip lz_search.hip"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet
dev = torch.device("cuda:0")
for name, N, half in (("b10c128", 16384, False), ("b6c64", 4096, False), ("b6c64", 2048, True)):
    torch.manual_seed(20260314)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev), half_workgroups=half)
    g = torch.Generator(device=dev).manual_seed(1)
    packed = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    packed[:, 0] = torch.randint(0, 1 << 36, (N,), device=dev, generator=g) | (1 << 50)
    packed[:, 1] = torch.randint(0, 1 << 36, (N,), device=dev, generator=g) & ~packed[:, 0] & ((1 << 36) - 1)
    out = f.forward_packed(packed)
    torch.cuda.synchronize()
    h = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in (out[0], out[1], out[2], out[4]))).hexdigest()[:16]
    k = 10 if N >= 16384 else 100
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(k):
            f.forward_packed(packed)
        torch.cuda.synchronize(); n += k
    dt = time.perf_counter() - t0
    print(f"  {name} N={N} half_wg={half}: {dt / n * 1e6:.1f} us per launch, {n * N / dt / 1e6:.3f} M evals/s = "
          f"{n * N / dt * f.flops_per_eval / 1e12:.0f} TFLOP/s, outputs sha256 {h}", flush=True)
''' % ROOT
variants = [("regular build", None)]
for tag in ("APF1", "APF2"):
    lib = os.path.join(ROOT, "liuzhou_amd", "_exp", f"liblz_{tag}.so")
    if os.path.exists(lib):
        variants.append((tag, lib))
variants.append(("regular build again", None))
for tag, lib in variants:
    env = dict(os.environ)
    if lib:
        env["LZ_HIP_LIB"] = lib
    print(tag, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env)
