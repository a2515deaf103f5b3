"""Exit-time lifetime of device objects (-m gpu).  Round 2's logs show a process that aborted AFTER pytest's summary
(`std::bad_variant_access`, `dumped core`): device objects kept alive by a failed test's traceback were finalised
during interpreter shutdown - or never - and HIP / RCCL were entered after, or torn down around, them.  By construction
now (`_native` "lifetime"; `tp_destroy` in include/tangency_posterior.h): ONE atexit hook closes every live Device
(batches first, then the communicator, streams, handle) before module teardown, finalisers do nothing once the
interpreter is finalising, and the library's own exit handler takes down what is left before the runtimes unload."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_exit_worker.py")


def _child(mode):
    return subprocess.run([sys.executable, WORKER, mode], capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("mode,code", [("raise", 1), ("exit3", 3), ("cycle", 1), ("global", 1), ("pinned", 1), ("clean", 0)])
def test_process_with_live_device_objects_exits_with_pythons_code(mode, code):
    """A Device with a one-rank RCCL communicator, an un-closed batch with a requested-but-unissued gather (or a pending
    asynchronous upload from pinned memory), then an exception / sys.exit: the child's exit status is Python's, not a
    signal, and stderr carries the traceback, not `terminate called`."""
    r = _child(mode)
    assert r.returncode == code, (r.returncode, r.stderr[-2000:])
    assert "terminate called" not in r.stderr and "core dumped" not in r.stderr, r.stderr[-2000:]
    if code == 1:
        assert "RuntimeError: boom" in r.stderr
