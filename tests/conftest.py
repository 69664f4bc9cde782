import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.gs_oracle import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def oracle64():
    from oracle.gs_oracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def emu_lib_path():
    """Host-emulated build of the kernel sources (tests/hipemu) -- a debugging aid, CPU only."""
    import subprocess
    d = os.path.join(ROOT, "tests", "hipemu")
    if os.environ.get("GS_EMU_LIB"):          # another build of the same sources (scripts/exp/emu_asan.sh: AddressSanitizer + UBSan)
        return os.path.abspath(os.environ["GS_EMU_LIB"])
    subprocess.check_call(["make", "-C", d, "-j8"], stdout=subprocess.DEVNULL)
    return os.path.join(d, "libgsplat_emu.so")


@pytest.fixture()
def emu(emu_lib_path):
    """Bind activesplat_amd to the emulated kernels for the duration of one test."""
    from tests import util
    undo = util.use_emulated_kernels(emu_lib_path)   # (the product path refuses host tensors; the emulated build is the one thing that takes them)
    yield "cpu"
    undo()


@pytest.fixture()
def hip():
    """Bind activesplat_amd to the real HIP library on cuda:0 (GPU tests)."""
    import torch
    from activesplat_amd import _lib
    _lib.unload_for_tests()
    assert torch.cuda.is_available(), "GPU test without a GPU"
    _lib.get()
    return "cuda"


def pytest_terminal_summary(terminalreporter):
    """How often check_backward's fp32 escape hatch decided a gradient comparison (tests/parity_cases.py)."""
    try:
        from tests import parity_cases as pc
    except Exception:
        return
    h = pc.HATCH
    if h["keys_checked"]:
        terminalreporter.write_line(f"check_backward: {h['keys_checked']} gradient tensors compared with the fp64 oracle at the stated tolerance; "
                                    f"fp32 escape hatch fired {h['fired']} time(s)" + (": " + ", ".join(f"{k} rel={r:.2e} frac={f:.4f} P={n}" for k, r, f, n in h["where"][:12]) if h["fired"] else "")
                                    + f"; decision-matched comparison (oracles re-run with the other decision at a proven alpha = 1/255 threshold pixel) decided {h['decisions']} time(s)"
                                    + (": " + ", ".join(f"{k} gaussians {[i for i, _, _, _ in w]}" for k, w, _, _ in h["decision_where"][:8]) if h["decisions"] else ""))
