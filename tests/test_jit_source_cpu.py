"""The device sources the engine hands to hiprtc at run time (csrc/ca_jit.cpp) must compile for gfx950: checked
here with hiprtc itself, which needs no GPU. The engine's own JIT path is exercised on the GPU by
tests/test_gpu_ca_parity.py::test_vn_truth_table_kernel_random_tables."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cellularautomatons3d_amd", "csrc")

PROGRAM = b"""
namespace ca3d_jit
{
#include "ca_bitops.inc"
#include "ca_packed_vn_kernel.inc"
}
"""


def _hiprtc():
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("hiprtc is not installed")


@pytest.mark.parametrize("cvl,lut_s,lut_b", [(2, 0x2A, 0x14), (1, 0xFF, 0x0A), (6, 0x00, 0x7E)])
def test_vn_kernel_source_compiles_with_hiprtc(cvl, lut_s, lut_b):
    rtc = _hiprtc()
    names = [b"ca_bitops.inc", b"ca_packed_vn_kernel.inc"]
    sources = [open(os.path.join(CSRC, n.decode()), "rb").read() for n in names]
    prog = C.c_void_p()
    hs = (C.c_char_p * 2)(*sources)
    hn = (C.c_char_p * 2)(*names)
    assert rtc.hiprtcCreateProgram(C.byref(prog), PROGRAM, b"ca3d_jit_vn.hip", 2, hs, hn) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-DCA3D_JIT=1", b"-DCA3D_JIT_CVL=%d" % cvl,
            b"-DCA3D_JIT_LS=%d" % lut_s, b"-DCA3D_JIT_LB=%d" % lut_b]
    rc = rtc.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*opts))
    n = C.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    log = C.create_string_buffer(n.value + 1)
    rtc.hiprtcGetProgramLog(prog, log)
    assert rc == 0, log.value.decode(errors="replace")
    size = C.c_size_t()
    assert rtc.hiprtcGetCodeSize(prog, C.byref(size)) == 0 and size.value > 1000
    code = C.create_string_buffer(size.value)
    assert rtc.hiprtcGetCode(prog, code) == 0
    assert b"ca3d_jit_vn_zr1" in code.raw and b"ca3d_jit_vn_zr2" in code.raw
    rtc.hiprtcDestroyProgram(C.byref(prog))
