"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/ca3d.h declares, and refuses to
run without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from cellularautomatons3d_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ca3d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ca3d_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(n for n, _, _ in _capi.SYMBOLS) == declared


def test_abi_version():
    assert _capi.load().ca3d_abi_version() == 7


def test_no_exception_crosses_the_c_abi():
    """include/ca3d.h promises status codes: a C++ exception leaving an extern "C" function ends a Node.js / ctypes host.
    (a) the boundary itself, run on the CPU through the test hook: std::bad_alloc -> CA3D_ERR_OUT_OF_MEMORY, anything else ->
    CA3D_ERR_DEVICE, message in ca3d_last_error(), and the process is still here; (b) every function the header declares is DEFINED as
    a function-try-block with that handler (CA3D_API_TRY ... CA3D_API_CATCH), the two trivial getters excepted."""
    lib = _capi.load()
    assert lib.ca3d_selftest_exception(0) == -4 and b"bad_alloc" in lib.ca3d_last_error()
    assert lib.ca3d_selftest_exception(1) == -3 and b"selftest" in lib.ca3d_last_error()
    assert lib.ca3d_selftest_exception(2) == -3 and b"unknown C++ exception" in lib.ca3d_last_error()
    assert lib.ca3d_selftest_exception(3) in (-3, -4) and lib.ca3d_last_error()  # a real oversized allocation
    assert lib.ca3d_selftest_exception(9) == 0
    src = ""
    for f in ("ca3d_api.cpp", "ca3d_group.cpp"):
        src += open(os.path.join(ROOT, "cellularautomatons3d_amd", "csrc", f)).read()
    trivial = {"ca3d_abi_version", "ca3d_last_error"}  # return a constant / a pointer to a static buffer: nothing to throw
    for name in _declared_symbols():
        if name in trivial:
            continue
        m = re.search(r"^int " + name + r"\([^;{]*?\) CA3D_API_TRY\n\{\n.*?\n\}\nCA3D_API_CATCH\n", src, flags=re.S | re.M)
        assert m, f"{name} is not defined behind the exception boundary"
        assert "CA3D_API_TRY" not in m.group(0)[m.group(0).index("CA3D_API_TRY") + 12:], name  # the match is ONE function


def test_no_cpu_fallback_when_no_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load()
    h = C.c_void_p()
    rc = lib.ca3d_create(0, C.byref(h))
    assert rc == -3 and not h.value
    assert b"no CPU fallback" in lib.ca3d_last_error()
    with pytest.raises(_capi.Ca3dError):
        from cellularautomatons3d_amd import Engine

        Engine(0)


def test_product_does_not_reference_oracle():
    # The oracle is test infrastructure: nothing in the package or the C sources may import, link or call it.
    pkg = os.path.join(ROOT, "cellularautomatons3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".js", ".c", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "libca3d_oracle" not in txt and "oracle_lib" not in txt and "ca3d_oracle_" not in txt, f


def test_every_option_is_documented_in_the_header():
    """ca3d_set_option / ca3d_group_set_option names are strings, not symbols: every name the engine compares against appears (quoted) in
    include/ca3d.h, so a host author finds it there and not only in the source."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "ca3d.h")).read()
    for src in ("ca3d_api.cpp", "ca3d_group.cpp"):
        text = open(os.path.join(root, "cellularautomatons3d_amd", "csrc", src)).read()
        names = set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', text))
        assert names, src
        missing = sorted(n for n in names if f'"{n}"' not in header)
        assert not missing, f"{src}: options not documented in include/ca3d.h: {missing}"
