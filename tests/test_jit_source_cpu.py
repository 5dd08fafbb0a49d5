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


CLASS_PROGRAM = b"""
#include "ca_device_types.h"
namespace ca3d
{
namespace jit
{
#include "ca_bitops.inc"
#include "ca_jit_rule.inc"
#include "ca_bitslice.inc"
#include "ca_packed_class_kernel.inc"
}
}
"""

ROLL_PROGRAM = b"""
#include "ca_device_types.h"
namespace ca3d
{
namespace jit
{
#include "ca_bitops.inc"
#include "ca_jit_rule.inc"
#include "ca_bitslice.inc"
#include "ca_packed_roll_kernel.inc"
}
}
"""

RESIDENT_PROGRAM = b"""
namespace ca3d_jit
{
#include "ca_bitops.inc"
#include "ca_resident_kernel.inc"
}
"""

RESIDENT_CLASS_PROGRAM = b"""
#include "ca_device_types.h"
namespace ca3d
{
namespace jit
{
#include "ca_bitops.inc"
#include "ca_jit_rule.inc"
#include "ca_bitslice.inc"
#include "ca_packed_roll_kernel.inc"
#include "ca_resident_kernel.inc"
#include "ca_resident_class_kernel.inc"
}
}
"""

ROWS_PROGRAM = b"""
#include "ca_device_types.h"
namespace ca3d
{
namespace jit
{
#include "ca_bitops.inc"
#include "ca_jit_rule.inc"
#include "ca_bitslice.inc"
#include "ca_packed_rows_kernel.inc"
}
}
"""

HEADERS = [b"ca_bitops.inc", b"ca_packed_vn_kernel.inc", b"ca_device_types.h", b"ca_bitslice.inc", b"ca_packed_class_kernel.inc",
           b"ca_packed_roll_kernel.inc", b"ca_resident_kernel.inc", b"ca_resident_class_kernel.inc", b"ca_packed_rows_kernel.inc"]


#: the clustered rule as rule_synth.cpp writes it (the generated header the engine passes as "ca_jit_rule.inc")
CLUSTERED_RULE_FN = b"""
#define CA3D_JIT_RULE_FN 1
__device__ __forceinline__ u32 jit_rule_word(u32 alive, const u32 *mn, const u32 *ed, const u32 *co)
{
	const u32 t14 = bitop3<0x02>(mn[4], mn[3], mn[2]);
	const u32 t15 = bitop3<0x38>(ed[2], ed[1], ed[0]);
	const u32 t16 = bitop3<0x44>(ed[3], t15, ed[3]);
	const u32 t17 = bitop3<0x1C>(co[2], co[1], co[0]);
	const u32 t18 = bitop3<0xFE>(t14, t16, t17);
	const u32 t19 = bitop3<0xE0>(mn[2], mn[1], mn[0]);
	const u32 t20 = bitop3<0x10>(t19, mn[4], mn[3]);
	const u32 t21 = bitop3<0x10>(ed[2], ed[1], ed[0]);
	const u32 t22 = bitop3<0x44>(ed[3], t21, ed[3]);
	const u32 t23 = bitop3<0x08>(co[2], co[1], co[0]);
	const u32 t24 = bitop3<0xFE>(t20, t22, t23);
	return bitop3<0xCA>(alive, t18, t24);
}
"""


def _compile(rtc, program, name, defines, rule_fn=b"// no synthesised rule\n"):
    names = HEADERS + [b"ca_jit_rule.inc"]
    sources = [open(os.path.join(CSRC, n.decode()), "rb").read() for n in HEADERS] + [rule_fn]
    prog = C.c_void_p()
    hs = (C.c_char_p * len(names))(*sources)
    hn = (C.c_char_p * len(names))(*names)
    assert rtc.hiprtcCreateProgram(C.byref(prog), program, name, len(names), hs, hn) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-DCA3D_JIT=1"] + defines
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
    rtc.hiprtcDestroyProgram(C.byref(prog))
    return code.raw


@pytest.mark.parametrize("main,e,c,zr,tables", [
    (2, "true", "true", 4, (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)),   # the clustered rule's shape
    (3, "false", "false", 2, (0x000C, 0x0008, 0, 0, 0, 0)),                              # Moore 2D, life
    (4, "false", "true", 4, (0x0350, 0x0244, 0, 0, 0x0002, 0x0000)),                     # edges main + corners set
])
def test_class_kernel_source_compiles_with_hiprtc(main, e, c, zr, tables):
    rtc = _hiprtc()
    defines = [b"-DCA3D_JIT_MAIN=%d" % main, b"-DCA3D_JIT_E=" + e.encode(), b"-DCA3D_JIT_C=" + c.encode(), b"-DCA3D_JIT_ZR=%d" % zr]
    defines += [b"-DCA3D_JIT_%s=%du" % (n, t) for n, t in zip([b"TS0", b"TB0", b"TS1", b"TB1", b"TS2", b"TB2"], tables)]
    code = _compile(rtc, CLASS_PROGRAM, b"ca3d_jit_class.hip", defines)
    assert b"ca3d_jit_class_deep" in code and b"ca3d_jit_class_deep_za" in code and b"ca3d_jit_class_flat" in code


@pytest.mark.parametrize("G,main,e,c,zr,tables", [
    (992, 0, "false", "false", 2, (0x7F, 0x0A, 0, 0, 0, 0)),                                 # "1000" in the UI, the start-up rule
    (96, 2, "true", "true", 4, (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)),   # rows of 3 words, the clustered rule's shape
    (32, 3, "false", "false", 2, (0x000C, 0x0008, 0, 0, 0, 0)),                              # one word per row
    (1984, 5, "true", "false", 4, (0x0014, 0x0008, 0x0038, 0x0010, 0, 0)),                   # rows of 62 words: one row per wave
])
def test_rows_kernel_source_compiles_with_hiprtc(G, main, e, c, zr, tables):
    """ca_packed_rows_kernel.inc: the kernel of every grid that is not a power of two (any multiple of 32)."""
    rtc = _hiprtc()
    defines = [b"-DCA3D_JIT_G=%d" % G, b"-DCA3D_JIT_MAIN=%d" % main, b"-DCA3D_JIT_E=" + e.encode(), b"-DCA3D_JIT_C=" + c.encode(), b"-DCA3D_JIT_ZR=%d" % zr]
    defines += [b"-DCA3D_JIT_%s=%du" % (n, t) for n, t in zip([b"TS0", b"TB0", b"TS1", b"TB1", b"TS2", b"TB2"], tables)]
    code = _compile(rtc, ROWS_PROGRAM, b"ca3d_jit_rows.hip", defines)
    assert b"ca3d_jit_rows_deep" in code and b"ca3d_jit_rows_flat" in code


@pytest.mark.parametrize("cvl,lut_s,lut_b", [(2, 0x2A, 0x14), (1, 0xFF, 0x0A), (6, 0x00, 0x7E)])
def test_vn_kernel_source_compiles_with_hiprtc(cvl, lut_s, lut_b):
    rtc = _hiprtc()
    code = _compile(rtc, PROGRAM, b"ca3d_jit_vn.hip", [b"-DCA3D_JIT_CVL=%d" % cvl, b"-DCA3D_JIT_LS=%d" % lut_s, b"-DCA3D_JIT_LB=%d" % lut_b])
    assert b"ca3d_jit_vn_zr1" in code and b"ca3d_jit_vn_zr2" in code


@pytest.mark.parametrize("cvl,main,e,c,tables", [
    (2, 2, "true", "true", (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)),   # 512^3, the clustered rule's shape
    (1, 3, "false", "false", (0x000C, 0x0008, 0, 0, 0, 0)),                              # 256^3, Moore 2D, life
    (6, 4, "false", "true", (0x0350, 0x0244, 0, 0, 0x0002, 0x0000)),                     # 8192^3, edges main + corners set
])
def test_roll_kernel_source_compiles_with_hiprtc(cvl, main, e, c, tables):
    rtc = _hiprtc()
    defines = [b"-DCA3D_JIT_CVL=%d" % cvl, b"-DCA3D_JIT_MAIN=%d" % main, b"-DCA3D_JIT_E=" + e.encode(), b"-DCA3D_JIT_C=" + c.encode()]
    defines += [b"-DCA3D_JIT_%s=%du" % (n, t) for n, t in zip([b"TS0", b"TB0", b"TS1", b"TB1", b"TS2", b"TB2"], tables)]
    code = _compile(rtc, ROLL_PROGRAM, b"ca3d_jit_roll.hip", defines, *([CLUSTERED_RULE_FN] if main == 2 and e == "true" else []))
    assert b"ca3d_jit_roll_z2" in code and b"ca3d_jit_roll_z4" in code and b"ca3d_jit_roll_z8" in code


@pytest.mark.parametrize("cv,main,e,c,tables", [
    (5, 2, "true", "true", (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)),   # 640^3, the clustered rule's shape
    (3, 4, "false", "true", (0x0350, 0x0244, 0, 0, 0x0002, 0x0000)),                     # 384^3, edges main + corners set
    (7, 0, "false", "false", (0x002A, 0x0014, 0, 0, 0, 0)),                              # 896^3, a von Neumann rule
])
def test_roll_np2_kernel_source_compiles_with_hiprtc(cv, main, e, c, tables):
    """The rolling-window kernel for rows of 3 / 5 / 6 / 7 uint4 (roll_step_np2): its own module, its own three entry points."""
    rtc = _hiprtc()
    defines = [b"-DCA3D_JIT_CV_NP2=%d" % cv, b"-DCA3D_JIT_MAIN=%d" % main, b"-DCA3D_JIT_E=" + e.encode(), b"-DCA3D_JIT_C=" + c.encode()]
    defines += [b"-DCA3D_JIT_%s=%du" % (n, t) for n, t in zip([b"TS0", b"TB0", b"TS1", b"TB1", b"TS2", b"TB2"], tables)]
    code = _compile(rtc, ROLL_PROGRAM, b"ca3d_jit_roll_np2.hip", defines, *([CLUSTERED_RULE_FN] if main == 2 and e == "true" else []))
    assert b"ca3d_jit_roll_np2_z2" in code and b"ca3d_jit_roll_np2_z4" in code and b"ca3d_jit_roll_np2_z8" in code
    assert b"ca3d_jit_roll_z8" not in code


def test_resident_kernel_source_compiles_with_hiprtc():
    code = _compile(_hiprtc(), RESIDENT_PROGRAM, b"ca3d_jit_resident.hip", [b"-DCA3D_JIT_LS=%d" % 0x2A, b"-DCA3D_JIT_LB=%d" % 0x14])
    assert b"ca3d_jit_resident" in code and b"ca3d_jit_resident_pair" in code and b"ca3d_jit_resident256" in code
    assert b"ca3d_jit_resident_stagger" not in code and b"ca3d_jit_resident256_deep" not in code  # removed in round 5 (lost twice)


@pytest.mark.parametrize("pz", [20, 24, 34, 36])
def test_resident_slab_kernel_fits_its_registers_and_lds(pz):
    """The slab form of the resident kernel for a rank's share of 1024^3 over eight (20 / 24 planes per tile layer) and over four
    ranks (34 / 36: the form that reads the rows beside a thread's own inside its main pass). A launch is only resident as a whole at
    four waves per SIMD (1024 threads per CU): no more than 128 registers, nothing spilled to scratch, the one LDS image inside
    the CU's 160 KB — read from the code object's metadata."""
    import re
    import subprocess
    code = _compile(_hiprtc(), RESIDENT_PROGRAM, b"ca3d_jit_resident_slab.hip", [b"-DCA3D_JIT_LS=255", b"-DCA3D_JIT_LB=10", b"-DCA3D_JIT_SLAB_PZ=%d" % pz])
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not in this image")
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ca3d_slab_%d.co" % pz)
    with open(path, "wb") as f:
        f.write(code)
    notes = subprocess.run([readelf, "--notes", path], capture_output=True, text=True, check=True).stdout
    os.unlink(path)
    blk = [b for b in notes.split(".agpr_count") if re.search(r"\.name:\s+ca3d_jit_resident_slab\b", b)]
    assert len(blk) == 1
    field = lambda name: int(re.search(r"\.%s:\s+(\d+)" % name, blk[0]).group(1))
    assert field("vgpr_count") <= 128 and field("vgpr_spill_count") == 0 and field("private_segment_fixed_size") == 0
    assert field("group_segment_fixed_size") <= 160 * 1024 and field("max_flat_workgroup_size") == 1024


def test_resident_class_kernel_source_compiles_with_hiprtc():
    """Both tile geometries (512^3: 16 words x 32 planes, 256^3: 8 x 8) of the resident class kernel, for the clustered rule."""
    main, e, c, tables = 2, "true", "true", (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)
    defines = [b"-DCA3D_JIT_MAIN=%d" % main, b"-DCA3D_JIT_E=" + e.encode(), b"-DCA3D_JIT_C=" + c.encode()]
    defines += [b"-DCA3D_JIT_%s=%du" % (n, t) for n, t in zip([b"TS0", b"TB0", b"TS1", b"TB1", b"TS2", b"TB2"], tables)]
    code = _compile(_hiprtc(), RESIDENT_CLASS_PROGRAM, b"ca3d_jit_resident_class.hip", defines, CLUSTERED_RULE_FN)
    assert b"ca3d_jit_resident_class256" in code and b"ca3d_jit_resident_class" in code
