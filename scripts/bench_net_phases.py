import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    f = FusedNet(ChessNet(**MODEL_CONFIGS[sys.argv[2]]).eval().to(dev))
    N = 4096 if sys.argv[2] == "b6c64" else 2048
    x = (torch.rand(N, 11, 6, 6, device=dev) < 0.3).float()
    for _ in range(5): f(x, want_logits=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f(x, want_logits=False)
    e.record(); torch.cuda.synchronize()
    print(f"{sys.argv[2]} stop={os.environ.get('LZ_NET_DEBUG_STOP','0')}: {s.elapsed_time(e) / 20 * 1000:.1f} us", flush=True)
else:
    for model in ("b6c64", "b10c128"):
        for stop in (1, 2, 3, 4, 5, 0):
            env = dict(os.environ, LZ_NET_DEBUG_STOP=str(stop))
            subprocess.run([sys.executable, __file__, "child", model], env=env, check=True)
