"""Child processes for everything in the GPU suite that touches an external runtime (RCCL, torch.distributed.run, bench.py).

Round 5's driver run ended at its second test: an in-process `init_process_group("nccl")` never returned and the in-process guard
(pytest-timeout, method "thread") ended the whole pytest process with it -- 214 parity tests unrun.  Since round 6 such calls live in
a child with a wall-clock limit of its own: a child that hangs is killed (its whole process group), the ONE test that started it
fails with the child's output and its RCCL log in the assertion message, and the suite goes on.

    run_child([...argv...], limit_s, env=...) -> ChildResult(rc, out, err, seconds, timed_out, rccl_log)
"""
from __future__ import annotations

import os
import signal
import subprocess
import sys
import tempfile
import time
from dataclasses import dataclass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RCCL_CHILD = os.path.join(ROOT, "tests", "rccl_child.py")


@dataclass
class ChildResult:
    rc: int
    out: str
    err: str
    seconds: float
    timed_out: bool
    rccl_log: str

    def report(self, tail: int = 3000) -> str:
        how = f"killed after its limit ({self.seconds:.0f} s)" if self.timed_out else f"exit code {self.rc} after {self.seconds:.1f} s"
        return (f"child {how}\n--- stdout (tail) ---\n{self.out[-tail:]}\n--- stderr (tail) ---\n{self.err[-tail:]}\n"
                f"--- RCCL log (tail) ---\n{self.rccl_log[-tail:]}")

    @property
    def ok(self) -> bool:
        return self.rc == 0 and not self.timed_out


def run_child(argv, limit_s: float, env=None, cwd=ROOT, rccl_debug: bool = False) -> ChildResult:
    """Run argv in a session of its own; SIGKILL to the whole process group when limit_s is over (torch.distributed.run forks
    workers: killing the launcher alone would leave them on the GPU).  Never raises on a hang -- the caller asserts on `.ok`."""
    env = dict(os.environ if env is None else env)
    log_path = None
    if rccl_debug:
        fd, log_path = tempfile.mkstemp(prefix="gf2bv_rccl_", suffix=".log")
        os.close(fd)
        env.setdefault("NCCL_DEBUG", "INFO")
        env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,BOOTSTRAP,ENV,NET")
        env["NCCL_DEBUG_FILE"] = log_path
    t0 = time.monotonic()
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:
        p = subprocess.Popen(list(argv), stdout=fo, stderr=fe, env=env, cwd=cwd, start_new_session=True)
        timed_out = False
        try:
            rc = p.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            rc = p.wait()
        fo.seek(0)
        fe.seek(0)
        out, err = fo.read(), fe.read()
    log = ""
    if log_path:
        try:
            log = open(log_path, errors="replace").read()
        except OSError:
            pass
        try:
            os.unlink(log_path)
        except OSError:
            pass
    return ChildResult(rc, out, err, time.monotonic() - t0, timed_out, log)


def rccl_child(mode: str, *extra: str, limit_s: float = 150, env=None) -> ChildResult:
    return run_child([sys.executable, RCCL_CHILD, mode, *extra], limit_s, env=env, rccl_debug=True)


_control = None


def rccl_control() -> ChildResult:
    """Is RCCL usable on this box AT ALL?  World-1 init + one all_gather from a process that never loads libgf2bv_hip.so (torch
    only).  Cached per pytest process.  The RCCL tests of the product run whatever this says; where they fail, its verdict goes
    into the message, so a record can tell 'RCCL does not come up on this box' from 'RCCL does not come up behind this library'."""
    global _control
    if _control is None:
        _control = rccl_child("control", limit_s=120)
    return _control
