"""Host-side latency of small solves (where does the time go?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import hip, m4ri_solve
from oracle import gf2_oracle as O
import random
rng = random.Random(1)
for rows, cols in ((4, 4), (640, 256), (640, 256), (640, 256), (2100, 2048), (2100, 2048)):
    eqs = [rng.getrandbits(cols + 1) for _ in range(rows)]
    aug = O.eqs_to_aug(eqs, cols)
    for mode in (0, 1, 0, 1):
        t = time.time(); s = hip.solve_words(aug, rows, cols, mode); dt = time.time() - t
        print(rows, cols, mode, f"{dt*1e3:.2f}ms", {k: round(v, 3) for k, v in s.stats.items() if k.startswith("ms_")}, flush=True)
