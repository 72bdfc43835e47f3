"""Randomised differential test on the GPU box: random shapes / densities / rank caps / modes / k_update
configurations against the CPU oracle, through the C ABI.  usage: stress_parity.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gf2bv_amd import hip
from oracle import gf2_oracle as O
from tests.systems import random_system

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
configs = ["0", "1", "2", "3", "4", "5"]          # instances of k_update16 (GF2_TW = 2 builds)
knobs = [{}, {}, {"GF2BV_FAST": "0"}, {"GF2BV_OPTIMISTIC": "0"}, {"GF2BV_SELF_WAIT_US": "0"}, {"GF2BV_FLAG_SYNC": "0"},
         {"GF2BV_TWO_LEVEL": "2"}, {"GF2BV_TWO_LEVEL": "3"}, {"GF2BV_TWO_LEVEL": "4"}, {"GF2BV_TWO_LEVEL": "8"}, {"GF2BV_TWO_LEVEL": "2", "GF2BV_FLAG_SYNC": "0"},
         # round 4: systems that fit the LDS take the one-launch kernel unless GF2BV_SMALL=0; its input by copy or straight from pinned memory
         {"GF2BV_SMALL": "0"}, {"GF2BV_SMALL": "0"}, {"GF2BV_SMALL_ZC": "0"}, {"GF2BV_SMALL_ZC": "1"},
         # round 5: the sparse block search off; the next panel's tiles beside / before the outer pass
         {"GF2BV_SPARSE_FAST": "0"}, {"GF2BV_TWO_LEVEL": "2", "GF2BV_OUTER_SIDE": "0"}, {"GF2BV_TWO_LEVEL": "3", "GF2BV_OUTER_SIDE": "0", "GF2BV_FLAG_SYNC": "0"},
         # round 6: the floor without heuristics (no stream-pair probes, no XCD pinning, events instead of memory gates, no optimistic enqueue)
         {"GF2BV_PLAIN": "1"}, {"GF2BV_PLAIN": "1", "GF2BV_TWO_LEVEL": "2"}, {"GF2BV_PLAIN": "1", "GF2BV_TWO_LEVEL": "4"},
         {"GF2BV_TWO_LEVEL": "2"}, {"GF2BV_TWO_LEVEL": "3"}]
t0, n, worst = time.time(), 0, 0
while time.time() - t0 < budget:
    cols = rng.choice([rng.randint(1, 130), rng.randint(131, 700), rng.randint(700, 2600), 64 * rng.randint(1, 40), 256 * rng.randint(1, 10) + rng.choice([-1, 0, 1]),
                       rng.randint(2048, 4200), rng.randint(2048, 6200)])               # (>= 8 blocks: the optimistic enqueue, the sparse search and their resume paths)
    cols = max(cols, 1)
    rows = cols + rng.choice([0, 1, rng.randint(0, 64), rng.randint(0, cols), rng.randint(0, 3 * cols)])
    density = rng.choice([0.5, 0.5, 0.1, 0.02, 0.004, 0.0015])
    cap = rng.choice([None, None, rng.randint(1, cols), max(1, cols - rng.randint(0, 5))])
    consistent = rng.random() < 0.8
    zero_rows = rng.choice([0, 0, rng.randint(0, rows // 2)])
    mode = rng.randint(0, 1)
    os.environ["GF2BV_UPDATE"] = rng.choice(configs)
    for k in ("GF2BV_FAST", "GF2BV_OPTIMISTIC", "GF2BV_SELF_WAIT_US", "GF2BV_FLAG_SYNC", "GF2BV_TWO_LEVEL", "GF2BV_SMALL", "GF2BV_SMALL_ZC",
              "GF2BV_SPARSE_FAST", "GF2BV_OUTER_SIDE", "GF2BV_PLAIN"):
        os.environ.pop(k, None)
    os.environ.update(rng.choice(knobs))
    eqs = random_system(rng, rows, cols, density, cap, consistent, min(zero_rows, rows - 1))
    rng.shuffle(eqs)
    aug = O.eqs_to_aug(eqs, cols)
    want = O.solve_words(aug, rows, cols, mode)
    got = hip.solve_words(aug, rows, cols, mode)
    ok = got.status == want["status"] and got.rank == want["rank"]
    if ok and want["status"] == 0:
        ok = np.array_equal(got.origin, want["origin"])
        if ok and mode == 1:
            ok = got.dimension == want["dim"] and np.array_equal(got.basis.reshape(-1), np.asarray(want["basis"]).reshape(-1))
    if ok:
        ok = np.array_equal(got.pivots, want["pivcols"][: want["rank"]])
    n += 1
    worst = max(worst, rows)
    if not ok:
        print(f"MISMATCH rows={rows} cols={cols} density={density} cap={cap} consistent={consistent} zero_rows={zero_rows} mode={mode} cfg={os.environ['GF2BV_UPDATE']} "
              f"knobs={ {k: v for k, v in os.environ.items() if k.startswith('GF2BV_')} }")
        sys.exit(1)
print(f"{n} random systems identical to the oracle in {time.time() - t0:.1f} s (largest {worst} rows)")
