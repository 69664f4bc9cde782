"""The C++ autograd front-end of the drop-in call (csrc/torch_frontend.cpp): build recipe and loader.

build() compiles it with g++ (host code only: the kernels live in libgsplat_hip.so) into activesplat_amd/_gs_frontend.so, in-tree, linked against
the C ABI library next to it and against this interpreter's torch.  get() imports it (building it first if it is missing) -- and fails loudly
when that is not possible."""
from __future__ import annotations

import importlib.machinery
import importlib.util
import os
import subprocess
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "torch_frontend.cpp")
SO = os.path.join(_HERE, "_gs_frontend.so")
_mod = None


def _deps():
    return [SRC, os.path.join(_HERE, "..", "include", "gsplat_hip.h")]


def _stale() -> bool:
    """Is the built module missing or older than its sources?  (Sources absent -- a binary-only install -- count as up to date.)"""
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False) -> str:
    """Compile csrc/torch_frontend.cpp -> _gs_frontend.so.  Safe against concurrent first use (the ranks of a torchrun launch, the processes of the
    tests): one builder at a time under an exclusive file lock, the compiler writes a private temporary file, and the finished library appears
    under its name atomically (os.replace) -- a process that finds _gs_frontend.so finds a complete file."""
    import fcntl
    import torch
    from torch.utils import cpp_extension as ce
    if not force and not _stale():
        return SO
    with open(SO + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():                  # (another process built it while this one waited for the lock)
                return SO
            tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
            inc = [f"-I{p}" for p in ce.include_paths(device_type="cuda")] + [f"-I{sysconfig.get_paths()['include']}"]
            tmp = f"{SO}.{os.getpid()}.tmp"
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_gs_frontend", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                   f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wall", "-Wno-unused-function",
                   *inc, SRC, "-o", tmp, f"-L{_HERE}", "-l:libgsplat_hip.so", f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip",
                   "-ltorch_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,/opt/rocm/lib"]
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, SO)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


#: set when the module could not be built or loaded here: (reason); rasterizer._frontend_apply then takes the Python twin -- the SAME library calls in
#: the same order through ctypes, ~150 us more host time per forward + backward -- after one warning.  Never a CPU path: the twin needs libgsplat_hip.so too.
unavailable = None


def get():
    """The built extension module (built or rebuilt first when it is missing or older than its sources).  RuntimeError when that is impossible
    (no g++ / torch headers on this box) or when the module was linked against another C ABI version."""
    global _mod
    if _mod is None:
        if _stale():
            # host code only (g++, ~25 s): built in place on first use when build() has not run here
            try:
                build()
            except Exception as e:
                raise RuntimeError(f"{SO} is missing or stale and could not be built ({e}): run `python -c 'import __graft_entry__ as g; g.build()'`") from e
        import torch  # noqa: F401  (its libraries must be loaded first)
        from . import _lib
        _lib.get()                                         # libgsplat_hip.so is resolved through the rpath; load it explicitly for a clear error
        loader = importlib.machinery.ExtensionFileLoader("_gs_frontend", SO)
        spec = importlib.util.spec_from_loader("_gs_frontend", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        if int(mod.abi_version()) != _lib.ABI_VERSION:
            raise RuntimeError(f"{SO}: linked against C ABI {mod.abi_version()}, this package needs {_lib.ABI_VERSION} -- rebuild")
        _mod = mod
    return _mod


def get_or_none():
    """get(), or None -- with ONE warning per process -- when the front-end cannot be built or loaded here; the caller then runs the Python twin."""
    global unavailable
    if unavailable is not None:
        return None
    try:
        return get()
    except Exception as e:
        import warnings
        unavailable = str(e)
        warnings.warn(f"activesplat_amd: the C++ autograd front-end is unavailable ({e}); the drop-in call runs through the Python twin "
                      "(same HIP library calls, more host time per frame)", RuntimeWarning, stacklevel=3)
        return None
