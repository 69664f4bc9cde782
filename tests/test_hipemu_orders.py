"""The host emulator's fibre-order modes (tests/hipemu/hipemu.cpp, HIPEMU_ORDER) are the tests' stand-in for a race detector: between two rendezvous the
order in which a workgroup's threads run is not defined, so the suites are also run with the emulator executing them in reverse and in random order
(scripts/exp/emu_orders.sh).  This self-test shows that the modes have teeth: a kernel whose missing barrier the natural order 0, 1, 2, ... hides gives the
expected result there, a wrong one in the other orders -- and the right one in every order once the barrier is in."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu")

RUN = """
import ctypes, sys, numpy as np
lib = ctypes.CDLL(sys.argv[1])
out = np.zeros(256, np.int32)
lib.probe(int(sys.argv[2]), out.ctypes.data_as(ctypes.c_void_p))
print(int((out == ((np.arange(256) + 255) % 256) + 1).sum()))
"""


def test_fibre_order_modes_expose_a_missing_barrier(tmp_path):
    so = str(tmp_path / "libprobe.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-I", EMU, "-x", "c++", os.path.join(EMU, "selftest", "order_probe.cpp"),
                           os.path.join(EMU, "hipemu.cpp"), "-o", so])

    def right(order, barrier):
        env = dict(os.environ, OMP_NUM_THREADS="1")
        env.pop("HIPEMU_ORDER", None)
        if order:
            env["HIPEMU_ORDER"] = order
        return int(subprocess.check_output([sys.executable, "-c", RUN, so, str(barrier)], env=env).decode().strip())
    for order in (None, "reverse", "random:1", "random:7"):
        assert right(order, 1) == 256, order                      # with the barrier: every order gives the result
    assert right(None, 0) >= 255                                   # without it the natural order hides the race (thread 0 alone reads an unwritten slot)
    assert right("reverse", 0) <= 1                                # ... reverse order reads every slot before it is written
    assert 0 < right("random:1", 0) < 256 and 0 < right("random:7", 0) < 256
