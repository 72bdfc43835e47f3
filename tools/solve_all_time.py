"""solve_all vs solve_one at large N with a kernel basis of dimension d (default 65536^2, d = 12): back-substitution of
the d + 1 right-hand sides in parity groups, then the enumeration of all 2^d solutions (device-filled chunks + one
int export per element).  usage: solve_all_time.py [N] [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(7)
wt = (n + 1 + 63) // 64
aug = rng.integers(0, 1 << 63, size=(n, wt), dtype=np.int64).view(np.uint64) ^ \
      (rng.integers(0, 2, size=(n, wt), dtype=np.int64).view(np.uint64) << np.uint64(63))
aug[:, -1] = 0
# d dependent columns: column c_dst := column c_src (scattered over the matrix), then a planted right-hand side
for k in range(d):
    src, dst = 3 * k + 1, n - 1 - 977 * k
    bit = (aug[:, src >> 6] >> np.uint64(src & 63)) & np.uint64(1)
    aug[:, dst >> 6] = (aug[:, dst >> 6] & ~(np.uint64(1) << np.uint64(dst & 63))) | (bit << np.uint64(dst & 63))
x = rng.integers(0, 1 << 63, size=wt, dtype=np.int64).view(np.uint64)
x[-1] = 0
par = np.bitwise_count(aug & x[None, :]).sum(axis=1) & 1
aug[:, n >> 6] |= par.astype(np.uint64) << np.uint64(n & 63)

stride = hip.padded_stride(n)
dev = np.zeros((n, stride), dtype=np.uint64)
dev[:, :wt] = aug
buf = hip.DeviceBuffer(dev.nbytes)
buf.upload(dev)
for mode in (0, 1, 0, 1):
    t = time.perf_counter()
    sol = hip.solve_device(buf.ptr, n, n, stride, mode)
    dt = time.perf_counter() - t
    print(f"mode {mode}: {dt * 1e3:8.2f} ms  rank {sol.rank} dim {sol.dimension}  eliminate {sol.stats['ms_eliminate']:.2f} "
          f"backsub {sol.stats['ms_backsub']:.2f} export {sol.stats['ms_export']:.2f}", flush=True)
assert hip.residual_device(buf.ptr, n, n, stride, sol.origin) == 0
dim = sol.dimension
cnt = 1 << min(dim, 16)
t = time.perf_counter()
elems = hip.space_enumerate(sol.origin, sol.basis, 0, cnt, gray=True)
t1 = time.perf_counter()
ints = [int.from_bytes(e.tobytes(), "little") for e in elems]
t2 = time.perf_counter()
print(f"enumerate {cnt} elements of {elems.shape[1]} words on the device: {(t1 - t) * 1e3:.2f} ms; "
      f"{cnt} int exports on the host: {(t2 - t1) * 1e3:.2f} ms")
bad = sum(hip.residual_device(buf.ptr, n, n, stride, elems[k]) for k in (0, 1, cnt // 2, cnt - 1))
print("residual of 4 sampled elements:", bad)
