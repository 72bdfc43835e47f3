"""First-light diagnostics on a GPU box: parity on a ladder of shapes, then timings."""
import os, sys, time, random, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
from gf2bv_amd import hip
from oracle import gf2_oracle as O
from tests.systems import random_system

print("devices", hip.device_count(), flush=True)
rng = random.Random(11)
fails = 0
cases = [(4,4,.5,None,True,1),(8,5,.5,None,True,0),(70,64,.5,None,True,0),(64,64,.5,None,True,0),(100,65,.5,None,True,0),
         (130,128,.5,None,True,0),(300,200,.5,40,True,0),(300,200,.1,None,True,20),(300,200,.5,40,False,0),
         (640,256,.05,None,True,0),(1000,1000,.5,None,True,0),(1100,1023,.5,900,True,0),(2100,2048,.02,None,True,50),
         (3000,2500,.5,None,True,0),(5000,4097,.5,4000,True,0)]
for (rows,cols,dens,cap,cons,zr) in cases:
    for mode in (0,1):
        eqs = random_system(rng, rows, cols, dens, cap, cons, zr)
        aug = O.eqs_to_aug(eqs, cols)
        want = O.solve_words(aug, rows, cols, mode)
        try:
            got = hip.solve_words(aug, rows, cols, mode)
            ok = got.status == want["status"] and got.rank == want["rank"] and np.array_equal(got.pivots, want["pivcols"])
            if ok and got.status == 0:
                ok = np.array_equal(got.origin, want["origin"])
                if mode == 1: ok = ok and np.array_equal(got.basis, want["basis"])
            print("OK " if ok else "BAD", rows, cols, dens, cap, cons, zr, "mode", mode, "rank", got.rank, want["rank"], "status", got.status, want["status"],
                  "ms", round(got.stats["ms_total"],2), flush=True)
            if not ok:
                fails += 1
                npv = min(len(got.pivots), len(want["pivcols"]))
                diff = np.nonzero(got.pivots[:npv] != want["pivcols"][:npv])[0]
                print("   first pivot diff", diff[:5], got.pivots[:8], want["pivcols"][:8])
                if got.status == 0 and want["status"] == 0:
                    print("   origin diff words", np.nonzero(got.origin != want["origin"])[0][:8])
        except Exception as e:
            fails += 1
            print("EXC", rows, cols, mode, repr(e)); traceback.print_exc()
print("fails", fails, flush=True)
try:
    g.smoke()
except Exception:
    traceback.print_exc()
# timings on device-resident synthetic systems
cfgs = os.environ.get("FL_CFGS", "3x16,4x16,3x14,2x12,2x14,2x16,1x10,1x12,1x16").split(",")
for n in [int(x) for x in os.environ.get("FL_SIZES", "8192,32768,65536").split(",")]:
    stride = hip.padded_stride(n)
    buf = hip.DeviceBuffer(n * stride * 8)
    for cfg in cfgs:
        os.environ["GF2BV_UPDATE"] = cfg
        hip.synth_device(buf.ptr, n, n, stride, 1234)
        t = time.time()
        sol = hip.solve_device(buf.ptr, n, n, stride, 0, time_kernels=True)
        dt = time.time() - t
        hip.synth_device(buf.ptr, n, n, stride, 1234)
        bad = hip.residual_device(buf.ptr, n, n, stride, sol.origin)
        s = sol.stats
        gbs = 16 * s["sweep_words"] / (s["ms_sweep"] * 1e-3) / 1e9 if s["ms_sweep"] else 0
        print(f"N={n} cfg={cfg} rank={sol.rank} bad={bad} wall={dt*1e3:.1f}ms elim={s['ms_eliminate']:.1f} sweep={s['ms_sweep']:.1f} back={s['ms_backsub']:.1f} sweepGB/s={gbs:.0f} rowxor/s={s['row_xors']/(s['ms_eliminate']*1e-3):.3e}", flush=True)
    if n <= 8192:
        aug = O.gen_synthetic(n, n, 1234)
        want = O.solve_words(aug, n, n, 0)
        print("  oracle parity", np.array_equal(want["origin"], sol.origin), want["rank"], sol.rank, flush=True)
    buf.free()
