import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gf2bv_amd import hip
from oracle import gf2_oracle as O
from tests.systems import random_system
def run(tag, shapes, trials, env):
    for k in ("GF2BV_SERIAL", "GF2BV_DEBUG_SYNC", "GF2BV_UPDATE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for (rows, cols) in shapes:
        res = ""
        for trial in trials:
            rng = random.Random(trial * 7 + rows)
            eqs = random_system(rng, rows, cols, .5, None, True, 0)
            aug = O.eqs_to_aug(eqs, cols)
            want = O.solve_words(aug, rows, cols, 0)
            for mode in (0, 1):
                got = hip.solve_words(aug, rows, cols, mode)
                ok = got.status == want["status"] and np.array_equal(got.origin, want["origin"])
                res += "." if ok else "X"
        print(tag, env, rows, cols, res, flush=True)
small = ((70, 64), (64, 64), (200, 128))
run("race?", small, range(6), {})
run("race?", small, range(6), {"GF2BV_DEBUG_SYNC": "1"})
run("race?", small, range(6), {"GF2BV_DEBUG_SYNC": "2"})
run("race?", small, range(6), {"GF2BV_DEBUG_SYNC": "3"})
for upd in ("3x13", "2x13", "1x13", "4x16"):
    run("bug2", ((640, 256),), (4, 5, 0), {"GF2BV_SERIAL": "1", "GF2BV_UPDATE": upd})
