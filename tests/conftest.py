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

__graft_entry__.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # a runaway test (e.g. enumerating a 2**40-point solution space) must die long before it can exhaust the
    # host; pytest-timeout is present in the image, the guard is skipped where it is not
    if not config.pluginmanager.hasplugin("timeout"):
        return
    import pytest
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(240, method="thread"))    # "thread": also ends a loop stuck in C
