"""Round-5 micro-measurement for the streaming worker: pinned allocation cost and whether it stalls launches of another
thread; D2H rate into pinned memory; torch.save rate from pinned memory with 1 / 2 / 4 writer threads."""
import os, sys, tempfile, threading, time, shutil
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
MB = 1 << 20

def pin(n_mb):
    t0 = time.perf_counter(); t = torch.empty(n_mb * MB, dtype=torch.uint8, pin_memory=True); return t, time.perf_counter() - t0

for n in (64, 256, 600):
    t, dt = pin(n); print(f"pinned alloc {n} MB: {dt*1e3:.1f} ms ({n/1024/dt:.2f} GB/s)"); del t
t0 = time.perf_counter(); t = torch.empty(600 * MB, dtype=torch.uint8).pin_memory(); print(f"empty().pin_memory() 600 MB: {(time.perf_counter()-t0)*1e3:.1f} ms"); del t
t, dt = pin(600); print(f"pinned alloc again 600 MB (cached by torch's host allocator?): {dt*1e3:.1f} ms")

# does a big pinned allocation in another thread stall kernel launches here?
stalls = []
def bg():
    a, d = pin(1200); stalls.append(("bg alloc 1200 MB", d)); del a
th = threading.Thread(target=bg); th.start()
mx = 0.0; n = 0; t_last = time.perf_counter()
while th.is_alive():
    x.add_(1); now = time.perf_counter(); mx = max(mx, now - t_last); t_last = now; n += 1
th.join(); print(stalls, f"max launch-to-launch gap in the main thread: {mx*1e3:.1f} ms over {n} launches")

# D2H into pinned
src = torch.empty(600 * MB, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); t.copy_(src, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"D2H 600 MB into pinned: {dt*1e3:.1f} ms ({0.6/dt:.1f} GB/s)")
# kernels while a D2H copy runs on a side stream
side = torch.cuda.Stream()
big = torch.zeros(64 * MB, device=dev)
def work(k=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): big.mul_(1.0001)
    torch.cuda.synchronize(); return time.perf_counter() - t0
a = work()
with torch.cuda.stream(side): t.copy_(src, non_blocking=True)
b = work()
print(f"50 x 256 MB elementwise kernels: {a*1e3:.1f} ms alone, {b*1e3:.1f} ms with a 600 MB D2H in flight")

tmp = tempfile.mkdtemp(prefix="lz_io_")
def save(i, mb=350):
    v = torch.from_numpy(t.numpy()[: mb * MB])
    t0 = time.perf_counter(); torch.save({"a": v}, os.path.join(tmp, f"f{i}.pt")); return time.perf_counter() - t0
d1 = save(0); print(f"torch.save 350 MB, 1 thread: {d1*1e3:.0f} ms ({0.35/d1:.2f} GB/s)")
d1 = save(0); print(f"torch.save 350 MB again: {d1*1e3:.0f} ms ({0.35/d1:.2f} GB/s)")
for k in (2, 4, 8):
    ths = [threading.Thread(target=save, args=(i,)) for i in range(k)]
    t0 = time.perf_counter(); [h.start() for h in ths]; [h.join() for h in ths]; dt = time.perf_counter() - t0
    print(f"torch.save 350 MB x {k} threads: {dt*1e3:.0f} ms ({0.35*k/dt:.2f} GB/s aggregate)")
try:
    torch.serialization.set_crc32_options(False)
    d1 = save(0); print(f"torch.save 350 MB without crc32: {d1*1e3:.0f} ms ({0.35/d1:.2f} GB/s)")
except Exception as e:
    print("no crc option", e)
# raw write speed
buf = t.numpy()[: 350 * MB]
t0 = time.perf_counter()
with open(os.path.join(tmp, "raw.bin"), "wb") as f: f.write(buf)
dt = time.perf_counter() - t0; print(f"raw write 350 MB: {dt*1e3:.0f} ms ({0.35/dt:.2f} GB/s)")
shutil.rmtree(tmp)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
