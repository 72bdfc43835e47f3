"""Randomised differential test of the BATCH path on the GPU box: random shapes / densities / rank caps / gang sizes (multiples of 8
take the XCD-pinned bulk update with streaming accesses, others the plain grid) / knobs, every member of every gang against the CPU
oracle, through gf2bv_solve_batch_device's host twin -- or (a third of the jobs) as lists of ints through m4ri_solve_many with a random chunk
size and device list (round 5: digits in chunks, gathered under the previous chunk's solve).  usage: stress_gangs.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gf2bv_amd import _internal, hip
from oracle import gf2_oracle as O
from tests.systems import random_system

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
knobs = [{}, {}, {}, {"GF2BV_XCD_PIN": "0"}, {"GF2BV_GANG_NT": "0"}, {"GF2BV_XCD_WGS": "5"}, {"GF2BV_XCD_WGS": "64"}, {"GF2BV_GANG_BS": "0"},
         {"GF2BV_FLAG_SYNC": "0"},
         {"GF2BV_BATCH_THREADS": "3"}, {"GF2BV_PLAIN": "1"}]
ALL = sorted({k for d in knobs for k in d} | {"GF2BV_GANG", "GF2BV_BATCH_CHUNK_MB"})
t0, n, nsys_total = time.time(), 0, 0
while time.time() - t0 < budget:
    cols = rng.choice([rng.randint(900, 1400), rng.randint(1400, 3000), 256 * rng.randint(4, 11) + rng.choice([-1, 0, 1]), rng.randint(2048, 3300)])
    rows = cols + rng.choice([0, 1, rng.randint(0, 64), rng.randint(0, cols // 2)])
    nsys = rng.choice([3, 8, 9, 16, 17, 24, 33])
    gang = rng.choice([8, 8, 16, 16, 24, 5, 32])
    mode = rng.randint(0, 1)
    for k in ALL:
        os.environ.pop(k, None)
    os.environ["GF2BV_GANG"] = str(gang)
    os.environ.update(rng.choice(knobs))
    kinds = [(None, .5, True), (rng.randint(1, cols), .5, True), (None, .02, True), (max(1, cols - rng.randint(0, 5)), .5, rng.random() < .5), (1, .5, True)]
    systems = []
    for s in range(nsys):
        cap, dens, cons = rng.choice(kinds)
        systems.append(random_system(rng, rows, cols, dens, cap, cons, rng.choice([0, 0, rng.randint(0, rows // 3)])))
    if rng.random() < .3:
        systems[rng.randrange(nsys)] = [0] * rows
    if rng.random() < .34:
        os.environ["GF2BV_BATCH_CHUNK_MB"] = str(rng.randint(1, 3))
        devs = rng.choice([None, [0], [0, 0], [0, 0, 0]])
        got = _internal.m4ri_solve_many(systems, cols, mode) if devs is None else _internal.m4ri_solve_many(systems, cols, mode, devs)
        for i, (eqs, g) in enumerate(zip(systems, got)):
            o = O.m4ri_solve(list(eqs), cols, mode)
            ok = (g == o) if (mode == 0 or o is None or g is None) else (g.dimension, g.origin, g.basis) == (o.dimension, o.origin, o.basis)
            if not ok:
                print(f"MISMATCH (m4ri_solve_many) system {i} of {nsys}: rows={rows} cols={cols} mode={mode} gang={gang} devices={devs} "
                      f"knobs={ {k: v for k, v in os.environ.items() if k.startswith('GF2BV_')} }")
                sys.exit(1)
        n += 1
        nsys_total += nsys
        continue
    augs = np.stack([O.eqs_to_aug(e, cols) for e in systems])
    got = hip.solve_batch_words(augs, rows, cols, mode)
    for i, (a, g) in enumerate(zip(augs, got)):
        w = O.solve_words(a, rows, cols, mode)
        ok = g.status == w["status"] and g.rank == w["rank"] and np.array_equal(g.pivots, w["pivcols"][: w["rank"]])
        if ok and w["status"] == 0:
            ok = np.array_equal(g.origin, w["origin"])
            if ok and mode == 1:
                ok = g.dimension == w["dim"] and np.array_equal(g.basis.reshape(-1), np.asarray(w["basis"]).reshape(-1))
        if not ok or g.stats["handover_retries"]:
            print(f"MISMATCH system {i} of {nsys}: rows={rows} cols={cols} mode={mode} gang={gang} rank {g.rank}/{w['rank']} retries {g.stats['handover_retries']} "
                  f"knobs={ {k: v for k, v in os.environ.items() if k.startswith('GF2BV_')} }")
            sys.exit(1)
    n += 1
    nsys_total += nsys
print(f"{n} random gangs jobs ({nsys_total} systems) identical to the oracle in {time.time() - t0:.1f} s")
