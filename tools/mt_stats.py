"""Device-side statistics of the MT19937 recovery systems (BASELINE configs[2]): phases, blocks taken by the one-launch search.
usage: mt_stats.py [bs ...]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import LinearSystem, hip
from tests.harness_models import MT19937
VAR = {32: None, 17: None, 9: None, 1: None, 1337: 19968 // 1337 + 10, 137: 19968 // 137 + 60}
for bs in [int(a) for a in sys.argv[1:]] or [32, 1, 1337]:
    rand = random.Random(3142)
    eff = ((bs - 1) & bs) or bs
    ns = 624 * 32 // eff if VAR[bs] is None else VAR[bs]
    obs = [rand.getrandbits(bs) for _ in range(ns)]
    lin = LinearSystem([32] * 624)
    mt = lin.gens()
    sym = MT19937(mt)
    zeros = [sym.getrandbits(bs) ^ o for o in obs] + [mt[0] ^ 0x80000000]
    eqs = lin.get_eqs(zeros)
    eqs += [0] * max(0, lin._cols - len(eqs))
    cols = lin._cols
    nd = (cols + 1 + 31) // 32
    mask = (1 << (32 * nd)) - 1
    dig = np.frombuffer(b"".join((abs(e) & mask).to_bytes(4 * nd, "little") for e in eqs), dtype=np.uint32)
    off = np.arange(len(eqs) + 1, dtype=np.int64) * nd
    dens = sum(bin(e).count("1") for e in eqs[:2000]) / 2000 / cols
    for rep in range(2):
        s = hip.solve_digits(dig, off, 32, len(eqs), cols, 0).stats
    print(f"bs={bs}: {len(eqs)} x {cols}, density {dens:.4f}: eliminate {s['ms_eliminate']:.2f} ms, backsub {s['ms_backsub']:.2f}, pack {s['ms_pack']:.2f}, "
          f"fast_blocks {s['fast_blocks']} of {(cols + 255) // 256}, handovers {s['search_handovers']}, sweeps {s['n_sweeps']}", flush=True)
