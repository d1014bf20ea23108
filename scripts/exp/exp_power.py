"""Power / clock held while the network kernel runs back to back (evidence for the power-limited regime).
Samples `rocm-smi` (average package power, sclk) from a thread while the fused network evaluates large batches."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
from liuzhou_amd.net_hip import FusedNet

samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True,
                                 timeout=10).stdout
            samples.append((time.time(), out.strip().replace("\n", " | ")[:400]))
        except Exception as exc:  # noqa
            samples.append((time.time(), f"rocm-smi failed: {exc!r}"))
        time.sleep(0.5)


dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "b6c64"
torch.manual_seed(20260314)
f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(dev))
x = (torch.rand(262144 if name == "b6c64" else 65536, 11, 6, 6, device=dev) < 0.3).float()
print("idle:", subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().replace("\n", " | ")[:400])
th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(20):
        f(x, want_logits=False)
    torch.cuda.synchronize()
    n += 20
dt = time.time() - t0
stop = True
th.join(timeout=2)
print(f"{name}: {n * x.shape[0] / dt / 1e6:.2f} M evals/s sustained over {dt:.1f} s")
for t, s in samples[:: max(1, len(samples) // 6)]:
    print(f"t={t - t0:5.1f}s {s}")
