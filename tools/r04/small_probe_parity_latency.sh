#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
GF2BV_SMALL_PROBE=1 python - > $O/r04_small_probe.txt 2>&1 <<'PY'
import random, sys
sys.path.insert(0, '.')
import numpy as np
from gf2bv_amd import hip
from tools.small_latency import eqs_to_aug
PY
python - > $O/r04_small_probe.txt 2>&1 <<'PY'
import os, random, sys
sys.path.insert(0, '.')
os.environ["GF2BV_SMALL_PROBE"] = "1"
import numpy as np
from gf2bv_amd import hip
def eqs_to_aug(eqs, cols):
    stride = (cols + 1 + 63) // 64
    mask = (1 << (cols + 1)) - 1
    buf = b"".join((((e & mask) >> 1) | ((e & 1) << cols)).to_bytes(stride * 8, "little") for e in eqs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(eqs), stride).copy()
rng = random.Random(1)
for rows, cols in ((4, 4), (128, 128), (640, 256), (800, 767)):
    eqs = [rng.getrandbits(cols + 1) for _ in range(rows)]
    aug = eqs_to_aug(eqs, cols)
    for k in range(3):
        hip.solve_words(aug, rows, cols, 0)
PY
timeout 600 python -m pytest tests/test_gpu_small.py -x -q -s > $O/r04_pytest10.log 2>&1; echo "small rc=$?" > $O/r04_gpu10.summary
{ echo "## GF2BV_SMALL=1"; timeout 120 python tools/small_latency.py; } > $O/r04_small_latency.txt 2>&1
