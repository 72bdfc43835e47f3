"""CPU cover of the harness pieces the GPU suite's robustness rests on (round 6): the child runner kills a hung child and its
descendants at the limit and reports instead of raising; the guarded process-group init ends a stuck rank with exit code 3; the
binaries carry the content hash of the sources next to them."""
import os
import sys
import textwrap
import time

from tests.child import ROOT, rccl_child, run_child


def test_hung_child_is_killed_with_its_descendants(tmp_path):
    res = rccl_child("hang", limit_s=3)
    assert res.timed_out and not res.ok and res.seconds < 20 and "hanging on purpose" in res.out
    # a child that forks a grandchild holding the output open (what torch.distributed.run and rocprofv3 do): the group goes
    script = tmp_path / "forker.py"
    script.write_text(textwrap.dedent("""
        import os, subprocess, sys, time
        p = subprocess.Popen([sys.executable, "-c", "import time; print('grandchild', flush=True); time.sleep(600)"])
        print("pid", p.pid, flush=True)
        time.sleep(600)
    """))
    t0 = time.monotonic()
    res = run_child([sys.executable, str(script)], 3)
    assert res.timed_out and time.monotonic() - t0 < 20
    pid = int([ln for ln in res.out.splitlines() if ln.startswith("pid")][0].split()[1])
    time.sleep(0.5)
    try:
        os.kill(pid, 0)
        alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
    except (ProcessLookupError, FileNotFoundError):
        alive = False
    assert not alive


def test_exit_code_and_output_are_reported():
    res = run_child([sys.executable, "-c", "import sys; print('out'); print('err', file=sys.stderr); sys.exit(7)"], 30)
    assert (res.rc, res.timed_out, res.ok) == (7, False, False) and res.out == "out\n" and res.err == "err\n"
    assert "exit code 7" in res.report()


def test_guarded_init_ends_a_stuck_rank(tmp_path):
    """world size 2 with one rank missing: gloo's rendezvous waits for the partner -- the watchdog ends the process with exit code 3
    after its limit and says where it was."""
    script = tmp_path / "stuck.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29333", RANK="1", WORLD_SIZE="2")
        from gf2bv_amd import batch
        batch.init_process_group_guarded("gloo", None, limit_s=4)
        print("came up?!")
    """))
    res = run_child([sys.executable, str(script)], 120)
    assert res.rc == 3 and not res.timed_out, res.report()
    assert "did not return within 4 s" in res.err and "init_process_group" in res.err and "came up" not in res.out


def test_binaries_carry_the_hash_of_their_sources():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "gf2bv_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    from gf2bv_amd import _internal, hip
    hid = b.source_id([os.path.join(b.CSRC, "gf2_solver.hip"), os.path.join(b.CSRC, "gf2_kernels.hip.h"),
                       os.path.join(ROOT, "include", "gf2bv_hip.h")])
    assert hip.build_id() == hid == b.binary_id(b.HIP_LIB, b.HIP_MARK) == _internal.build_id()["hip"]
    assert _internal.build_id()["shim"] == b.binary_id(b.EXT, b.EXT_MARK) and _internal.build_id()["shim"].endswith("-" + hid)
