import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Build the HIP library, the CPython shim and the CPU oracle BEFORE test modules are collected
# (they import gf2bv_amd at module level; the built artefacts are git-ignored).
import __graft_entry__  # noqa: E402

__graft_entry__.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
