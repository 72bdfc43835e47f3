"""Round 5: where a batch of MT19937 recovery systems spends its time -- the list-of-int conversion, the upload, or the solves.
Calls gf2bv_solve_batch_digits on digits prepared beforehand (32-bit digits cut from int.to_bytes) and prints the call's wall time
beside the per-system device statistics.  usage: mt_batch_digits_time.py [nsys] [bs]"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem, _internal, hip
from tests.harness_models import MT19937
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eff = ((bs - 1) & bs) or bs
systems = []
for s in range(nsys):
    rand = random.Random(3142 + s)
    obs = [rand.getrandbits(bs) for _ in range(624 * 32 // eff)]
    lin = LinearSystem([32] * 624)
    mt = lin.gens()
    sym = MT19937(mt)
    eqs = lin.get_eqs([sym.getrandbits(bs) ^ o for o in obs] + [mt[0] ^ 0x80000000])
    eqs += [0] * max(0, lin._cols - len(eqs))
    systems.append(eqs)
cols = lin._cols
rows = len(systems[0])
nw = (cols + 1 + 31) // 32
t0 = time.perf_counter()
digits = np.frombuffer(b"".join(e.to_bytes(nw * 4, "little") for eqs in systems for e in eqs), dtype=np.uint32)
offsets = np.arange(nsys * rows + 1, dtype=np.int64) * nw
t1 = time.perf_counter()
print(f"{nsys} systems {rows} x {cols}: digits {digits.nbytes / 1e6:.0f} MB, built in {1e3 * (t1 - t0):.0f} ms", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    sols = hip.solve_batch_digits(digits, offsets, 32, nsys, rows, cols, 0)
    t1 = time.perf_counter()
    st = [s.stats for s in sols]
    print(f"  call {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / nsys:.1f} per system); per system: gang {st[0]['gang_systems']}, eliminate "
          f"{np.mean([x['ms_eliminate'] for x in st]):.1f} ms, total {np.mean([x['ms_total'] for x in st]):.1f} ms, fast blocks "
          f"{st[0]['fast_blocks']}, hand-overs {st[0]['search_handovers']}, retries {sum(x['handover_retries'] for x in st)}", flush=True)
    t0 = time.perf_counter()
    one = [hip.solve_digits(digits[s * rows * nw:(s + 1) * rows * nw], offsets[:rows + 1], 32, rows, cols, 0) for s in range(nsys)]
    t1 = time.perf_counter()
    print(f"  one by one {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / nsys:.1f} per system), eliminate {np.mean([x.stats['ms_eliminate'] for x in one]):.1f}, total "
          f"{np.mean([x.stats['ms_total'] for x in one]):.1f}; same answers: {all(a.origin_int() == b.origin_int() for a, b in zip(sols, one))}; fast blocks {[x.stats['fast_blocks'] for x in one]}, "
          f"eliminate {[round(x.stats['ms_eliminate'], 1) for x in one]}, retries {sum(x.stats['handover_retries'] for x in one)}", flush=True)
t0 = time.perf_counter()
many = _internal.m4ri_solve_many(systems, cols, 0)
t1 = time.perf_counter()
print(f"  m4ri_solve_many (list of int) {1e3 * (t1 - t0):.1f} ms", flush=True)
