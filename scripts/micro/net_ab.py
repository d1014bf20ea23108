"""Same-box A/B of the fused network kernel between builds of the HIP library (LZ_HIP_LIB): one subprocess per library,
the same seeded weights and inputs, per-launch time with HIP events at the three launch shapes of the bench (C2 half
launch, C2 full launch, C3 launch) and a hash of all outputs (a bit-identical rewrite must keep it).
usage: python scripts/micro/net_ab.py libA.so libB.so ...   ("default" = the in-tree library)"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    sys.path.insert(0, ROOT)
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    dev = torch.device("cuda:0")
    out = {"lib": os.path.basename(os.environ.get("LZ_HIP_LIB", "default"))}
    for name, N, half in (("b6c64", 2048, True), ("b6c64", 4096, False), ("b10c128", 16384, False)):
        torch.manual_seed(20260314)
        f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev)).variant(half_workgroups=half)
        g = torch.Generator(device="cpu").manual_seed(5)
        x = (torch.rand(N, 11, 6, 6, generator=g) < 0.3).float().to(dev)
        for _ in range(10):
            o = f(x, want_logits=True)
        torch.cuda.synchronize()
        n = 300 if N <= 4096 else 40
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            o = f(x, want_logits=True)
        b.record(); torch.cuda.synchronize()
        h = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in list(o) + [f.last_value])).hexdigest()[:16]
        key = f"{name}@{N}{'h' if half else ''}"
        out["us " + key] = round(a.elapsed_time(b) / n * 1e3, 2)
        out["sha " + key] = h
        # the search loop's form: 32-byte packed states in, no value logits out
        packed = torch.zeros((N, 4), dtype=torch.int64, device=dev)
        packed[:, 0] = torch.randint(0, 1 << 36, (N,), generator=g).to(dev) | (1 << 50)
        packed[:, 1] = torch.randint(0, 1 << 36, (N,), generator=g).to(dev) & ~packed[:, 0] & ((1 << 36) - 1)
        for _ in range(10):
            o = f.forward_packed(packed)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            o = f.forward_packed(packed)
        b.record(); torch.cuda.synchronize()
        out["us packed " + key] = round(a.elapsed_time(b) / n * 1e3, 2)
        out["sha packed " + key] = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in o if t is not None)).hexdigest()[:16]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--child":
        child()
    else:
        for rep in range(2):
            for lib in sys.argv[1:]:
                env = dict(os.environ)
                if lib != "default":
                    env["LZ_HIP_LIB"] = os.path.join(ROOT, lib)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=True)
