"""Same-box A/B of the root bandit kernel between builds of the HIP library: one subprocess per library (LZ_HIP_LIB), the
same seeded roots (widths ~ the legal-move counts of a game: mean ~11.7, 81 % <= 16), the kernel timed alone with HIP events.
usage: python scripts/micro/bandit_ab.py libA.so libB.so ...      (paths relative to the repository root)"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from liuzhou_amd import v0_core
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    R, A = 16384, 64
    w = np.clip(np.round(rng.gamma(2.2, 5.3, size=R)), 1, 60).astype(np.int64)
    valid = torch.from_numpy(np.arange(A)[None, :] < w[:, None]).to(dev)
    pri = torch.from_numpy(rng.random((R, A), dtype=np.float32)).to(dev) * valid
    pri = pri / pri.sum(1, keepdim=True)
    leaf = torch.from_numpy(rng.uniform(-1, 1, (R, A)).astype(np.float32)).to(dev)
    out = {"lib": os.environ.get("LZ_HIP_LIB", "default"), "frac_le16": float((w <= 16).mean()), "mean_width": float(w.mean())}
    for sims in (1024, 8192, 65536):
        v = v0_core.root_puct_allocate_visits(pri, leaf, valid, sims, 1.5)        # warm-up (tables, scratch)
        torch.cuda.synchronize()
        n = 20 if sims <= 8192 else 5
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            v = v0_core.root_puct_allocate_visits(pri, leaf, valid, sims, 1.5)
        b.record(); torch.cuda.synchronize()
        h = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in v)).hexdigest()[:16]
        out[f"ms@{sims}"] = round(a.elapsed_time(b) / n, 4)
        out[f"sha@{sims}"] = h
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
