"""Host-side latency of small solves (where does the time go?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import hip
import random


def eqs_to_aug(eqs, cols):
    """equation ints (bit 0 = constant, bit k = variable k-1) -> packed augmented words, as INTEGRATION.md section 3"""
    stride = (cols + 1 + 63) // 64
    mask = (1 << (cols + 1)) - 1
    buf = b"".join((((e & mask) >> 1) | ((e & 1) << cols)).to_bytes(stride * 8, "little") for e in eqs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(eqs), stride).copy()


rng = random.Random(1)
for rows, cols in ((4, 4), (640, 256), (640, 256), (640, 256), (2100, 2048), (2100, 2048)):
    eqs = [rng.getrandbits(cols + 1) for _ in range(rows)]
    aug = eqs_to_aug(eqs, cols)
    for mode in (0, 1, 0, 1):
        t = time.time(); s = hip.solve_words(aug, rows, cols, mode); dt = time.time() - t
        print(rows, cols, mode, f"{dt*1e3:.2f}ms", {k: round(v, 3) for k, v in s.stats.items() if k.startswith("ms_")}, flush=True)
