"""Soak: many solves in flight at once (more than the runtime has hardware queues), mixed shapes and rank caps, every result
against the oracle.  The in-kernel hand-overs (k_block_fast_narrow's progress counter, k_prio_window's own gate, the stream
gates) are timing-dependent: this is where a missing write-through store showed (rank 2814 for 2600).
usage: soak_concurrent.py [rounds] [threads] [seed]"""
import os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gf2bv_amd import hip
from oracle import gf2_oracle as O
from tests.systems import random_system

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 2026)
shapes = [(2700, 2600, None), (5000, 4097, 2600), (2300, 2200, 2193), (9000, 1300, 700), (3100, 3000, 300), (4200, 4100, None),
          (6000, 5200, 5199), (1600, 1500, 1024), (7000, 6100, None), (3000, 2600, 2304)]
jobs = []
t0 = time.time()
for rows, cols, cap in shapes:
    aug = O.eqs_to_aug(random_system(rng, rows, cols, .5, cap, rng.random() < .8, 0), cols)
    jobs.append((aug, rows, cols, O.solve_words(aug, rows, cols, 1)))
print(f"{len(jobs)} systems, oracle {time.time() - t0:.1f} s", flush=True)
bad = 0
for r in range(rounds):
    order = jobs * 4
    rng.shuffle(order)
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda j: hip.solve_words(j[0], j[1], j[2], 1), order))
    for j, g in zip(order, res):
        w = j[3]
        ok = g.status == w["status"] and g.rank == w["rank"] and np.array_equal(g.pivots, w["pivcols"][: w["rank"]])
        if ok and w["status"] == 0:
            ok = np.array_equal(g.origin, w["origin"]) and np.array_equal(g.basis.reshape(-1), np.asarray(w["basis"]).reshape(-1))
        if not ok:
            bad += 1
            print(f"round {r}: MISMATCH rows={j[1]} cols={j[2]} rank {g.rank} want {w['rank']} status {g.status}/{w['status']}", flush=True)
    print(f"round {r}: {len(order)} solves, {bad} wrong so far, {time.time() - t0:.1f} s", flush=True)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
