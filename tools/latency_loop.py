import os, sys, time, random
sys.path.insert(0, "/root/repo")
import numpy as np
from gf2bv_amd import hip
rng = random.Random(1)
def eqs_to_aug(eqs, cols):
    stride = (cols + 1 + 63) // 64
    mask = (1 << (cols + 1)) - 1
    buf = b"".join((((e & mask) >> 1) | ((e & 1) << cols)).to_bytes(stride * 8, "little") for e in eqs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(eqs), stride).copy()
rows, cols = 2100, 2048
eqs = [rng.getrandbits(cols + 1) for _ in range(rows)]
aug = eqs_to_aug(eqs, cols)
ts = []
for i in range(300):
    t = time.time(); s = hip.solve_words(aug, rows, cols, i & 1); dt = (time.time() - t) * 1e3
    ts.append(dt)
    if dt > 5: print(i, f"{dt:.1f} ms", {k: round(v, 2) for k, v in s.stats.items() if k.startswith("ms_")}, flush=True)
ts = np.array(ts[5:])
print(f"median {np.median(ts):.2f} ms, p99 {np.percentile(ts, 99):.2f}, max {ts.max():.2f}, >5ms: {(ts > 5).sum()}")
