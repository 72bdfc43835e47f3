import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gf2bv_amd import hip
from oracle import gf2_oracle as O
rng = random.Random(1)
rows, cols = 2100, 2048
eqs = [rng.getrandbits(cols + 1) for _ in range(rows)]
aug = O.eqs_to_aug(eqs, cols)
for i in range(3):
    t = time.time(); s = hip.solve_words(aug, rows, cols, 0); print("wall", (time.time()-t)*1e3, file=sys.stderr)
