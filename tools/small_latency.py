"""Warm wall time of small solves through the three boundaries (median / p10 of 300 calls): _internal.m4ri_solve (list of
ints -> digits), hip.solve_words (packed host words), hip.solve_device (matrix resident on the device).
GF2BV_SMALL=0: the blocked multi-launch path instead of the one-launch kernel."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import _internal, hip


def eqs_to_aug(eqs, cols):
    """equation ints (bit 0 = constant, bit k = variable k-1) -> packed augmented words, as INTEGRATION.md section 3"""
    stride = (cols + 1 + 63) // 64
    mask = (1 << (cols + 1)) - 1
    buf = b"".join((((e & mask) >> 1) | ((e & 1) << cols)).to_bytes(stride * 8, "little") for e in eqs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(eqs), stride).copy()


def stat(fn, n=300):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts.sort()
    return f"median {ts[n // 2] * 1e6:7.1f} us  p10 {ts[n // 10] * 1e6:7.1f} us"


rng = random.Random(1)
SHAPES = ((4, 4, .5), (128, 128, .02), (640, 256, .05), (640, 256, .5), (1000, 1000, .5), (2100, 2048, .5))
if len(sys.argv) > 1:
    SHAPES = tuple((int(a.split("x")[0]), int(a.split("x")[1]), .5) for a in sys.argv[1:])
for rows, cols, dens in SHAPES:
    eqs = [rng.getrandbits(cols + 1) & rng.getrandbits(cols + 1) if dens < .1 else rng.getrandbits(cols + 1) for _ in range(rows)]
    aug = eqs_to_aug(eqs, cols)
    stride = hip.padded_stride(cols)
    wide = np.zeros((rows, stride), dtype=np.uint64); wide[:, :aug.shape[1]] = aug
    buf = hip.DeviceBuffer(wide.nbytes); buf.upload(wide)
    for mode in (0, 1):
        n = 300 if cols <= 1024 else 60
        print(f"{rows:5d} x {cols:4d} mode {mode}: m4ri_solve {stat(lambda: _internal.m4ri_solve(eqs, cols, mode), n)} | solve_words {stat(lambda: hip.solve_words(aug, rows, cols, mode), n)}"
              f" | solve_device {stat(lambda: hip.solve_device(buf.ptr, rows, cols, stride, mode), n)}", flush=True)
    buf.free()
