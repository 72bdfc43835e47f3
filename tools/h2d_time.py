import time, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gf2bv_amd import hip
n = 53 * 1024 * 1024
a = np.random.randint(0, 255, n, dtype=np.uint8)
buf = hip.DeviceBuffer(n)
for rep in range(3):
    t = time.perf_counter(); buf.upload(a); dt = time.perf_counter() - t
    print(f"pageable hipMemcpy {n/1e6:.0f} MB: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")
ta = torch.from_numpy(a)
tp = ta.pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); d.copy_(tp, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"pinned copy: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")
for rep in range(2):
    t = time.perf_counter(); b = np.empty_like(a); b[:] = a; dt = time.perf_counter() - t
    print(f"host memcpy: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")
    t = time.perf_counter(); z = np.zeros(n, dtype=np.uint8); z[0] = 1; dt = time.perf_counter() - t
    print(f"zero-fill alloc: {dt*1e3:.2f} ms")
