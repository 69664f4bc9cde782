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


def build(force: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension as ce
    deps = [SRC, os.path.join(_HERE, "..", "include", "gsplat_hip.h")]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [f"-I{p}" for p in ce.include_paths(device_type="cuda")] + [f"-I{sysconfig.get_paths()['include']}"]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_gs_frontend", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wall", "-Wno-unused-function",
           *inc, SRC, "-o", SO, f"-L{_HERE}", "-l:libgsplat_hip.so", f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip",
           "-ltorch_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return SO


def get():
    """The built extension module; RuntimeError when it is missing or was built against another C ABI version."""
    global _mod
    if _mod is None:
        if not os.path.exists(SO):
            # host code only (g++, ~25 s): built in place on first use when build() has not run here; a failing build raises -- there is no
            # silent fall-back to the slower Python twin on a GPU box
            try:
                build()
            except Exception as e:
                raise RuntimeError(f"{SO} is missing and could not be built ({e}): run `python -c 'import __graft_entry__ as g; g.build()'`") from e
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
