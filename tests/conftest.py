import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Build the HIP library, the CPython shim and the CPU oracle BEFORE test modules are collected
# (they import gf2bv_amd at module level; the built artefacts are git-ignored).
# torch FIRST: it brings its own copy of the HIP runtime, and a process must end up with ONE -- libgf2bv_hip.so then
# binds to the copy that is already loaded (as in bench.py); loaded the other way round, torch.cuda sees no device.
import torch  # noqa: E402,F401

import __graft_entry__  # noqa: E402

_BUILD = __graft_entry__.build()


# Order of the GPU suite (round 6): parity of the solve path first, everything that talks to an external runtime LAST.  Round 5's
# driver run died at its second test -- an RCCL init that never returned, in the alphabetically first file -- and 214 parity tests
# went unrun.  Within a file the order of definition is kept; tests marked `external` (RCCL, torch.distributed.run, bench.py
# launches: all of them run in child processes with limits of their own, tests/child.py) go behind every in-process test.
_FILE_ORDER = ("test_gpu_parity", "test_gpu_small", "test_gpu_space", "test_gpu_packed", "test_gpu_stress", "test_gpu_slab",
               "test_gpu_batch_c4", "test_gpu_external")


def _order_key(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    rank = _FILE_ORDER.index(name) if name in _FILE_ORDER else -1          # CPU files keep their place in front
    return (1 if item.get_closest_marker("external") else 0, rank)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "default_only: (test_gpu_parity.py) not repeated under GF2BV_PLAIN=1 -- asserts on a heuristic that mode turns off")
    config.addinivalue_line("markers", "external: talks to RCCL / launches torch.distributed.run or bench.py in a child process; "
                                       "ordered behind every in-process test")
    print(f"[gf2bv tests] libgf2bv_hip.so {_BUILD['hip'][0]} {_BUILD['hip'][1]}; _internal {_BUILD['shim'][0]} {_BUILD['shim'][1]}",
          file=sys.stderr, flush=True)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_order_key)                                            # (stable: definition order inside a file survives)
    # a runaway in-process test (e.g. enumerating a 2**40-point solution space, a kernel that spins) must die long before it can
    # exhaust the host.  "thread" ends the pytest process -- the only thing that stops a loop stuck in C -- so nothing that can hang
    # for reasons outside this library runs in-process any more (tests/child.py); pytest-timeout is present in the image, the guard
    # is skipped where it is not
    if not config.pluginmanager.hasplugin("timeout"):
        return
    import pytest
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(240, method="thread"))
