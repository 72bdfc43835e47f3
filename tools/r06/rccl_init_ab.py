"""is RCCL's world-1 init sensitive to what this library left on the device?  Each row is ONE fresh child
process (tests/rccl_child.py through tests/child.py: own limit, NCCL_DEBUG=INFO log captured); run it on several fresh leases and
append the tables to profiles/r06_rccl_init.txt.
usage (GPU box, repo root): python tools/r06/rccl_init_ab.py [label] >> gpurun_out/r06_rccl_init.txt"""
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.child import rccl_child  # noqa: E402

ROWS = [
    ("control: torch only, library never loaded", ("control",), {}),
    ("gather on a fresh device", ("gather",), {}),
    ("round 5's sequence: 64 x 32768^2 batch + single solve, then init", ("gather", "--after-solve", "64"), {}),
    ("same, GF2BV_STREAM_PAIRS=0 (no probed low-priority streams)", ("gather", "--after-solve", "64"), {"GF2BV_STREAM_PAIRS": "0"}),
    ("same, both pools trimmed + device synchronised before init", ("gather", "--after-solve", "64", "--trim"), {}),
    ("same, lazy init (no device_id: connect at the first collective)", ("gather", "--after-solve", "64", "--lazy"), {}),
    ("slab schedule over RCCL", ("slab",), {}),
]


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "lease"
    print(f"## {label}: host {socket.gethostname()}, {time.strftime('%Y-%m-%d %H:%M:%S')}", flush=True)
    for name, argv, env in ROWS:
        res = rccl_child(*argv, limit_s=300, env=dict(os.environ, **env))
        info = {}
        for ln in res.out.splitlines():
            if ln.startswith("CHILD_OK "):
                info = json.loads(ln[9:])
        verdict = "ok" if res.ok else ("HUNG (killed)" if res.timed_out else f"FAILED rc {res.rc}")
        print(f"{verdict:14s} init {info.get('init_seconds', '-'):>7} s  child {res.seconds:6.1f} s  | {name}", flush=True)
        if not res.ok:
            print(res.report(2500), flush=True)


if __name__ == "__main__":
    main()
