"""ctypes binding of libca3d.so (include/ca3d.h). Fails loudly if the library is missing: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libca3d.so")

CA3D_OK = 0
LAYOUT_PACKED32 = 0
LAYOUT_UNPACKED = 1
LUT_LEN = 81
COMM_ID_BYTES = 128

SLAB_SEND_LOW, SLAB_SEND_HIGH, SLAB_RECV_LOW, SLAB_RECV_HIGH, SLAB_OWNED = range(5)
SLAB_PHASE_ALL, SLAB_PHASE_EDGES, SLAB_PHASE_INTERIOR = range(3)


class Ca3dError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"ca3d error {code}: {message}")
        self.code = code
        self.message = message


class Info(C.Structure):
    _fields_ = [
        ("grid_size", C.c_uint32), ("layout", C.c_int32), ("z0", C.c_uint32), ("nz", C.c_uint32),
        ("ghost", C.c_uint32), ("step", C.c_uint64), ("state_words", C.c_uint64),
        ("current_buffer", C.c_int32), ("device", C.c_int32), ("kernel_name", C.c_char * 64),
        ("launches_total", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("steps", C.c_uint64), ("kernel_launches", C.c_uint64), ("gpu_ms", C.c_double),
        ("cell_steps", C.c_double), ("algorithmic_bytes", C.c_double),
    ]


class RenderStats(C.Structure):
    _fields_ = [
        ("gpu_ms", C.c_double), ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("primary_cell_visits", C.c_uint64), ("shadow_cell_visits", C.c_uint64),
    ]


class CommInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("comm_ranks", C.c_int32), ("comm_rank", C.c_int32), ("comm_device", C.c_int32),
                ("pci_bus_id", C.c_char * 32)]


class JitStats(C.Structure):
    _fields_ = [
        ("programs_compiled", C.c_uint64), ("programs_from_disk", C.c_uint64), ("programs_from_memory", C.c_uint64),
        ("compile_ms", C.c_double), ("disk_read_ms", C.c_double), ("load_ms", C.c_double), ("cache_dir", C.c_char * 256),
    ]


def jit_stats() -> dict:
    """The run-time compiler's counters for this process (include/ca3d.h ca3d_jit_stats) as a dict."""
    st = JitStats()
    check(load().ca3d_get_jit_stats(C.byref(st)))
    return {"programs_compiled": int(st.programs_compiled), "programs_from_disk": int(st.programs_from_disk),
            "programs_from_memory": int(st.programs_from_memory), "compile_ms": round(st.compile_ms, 2),
            "disk_read_ms": round(st.disk_read_ms, 2), "load_ms": round(st.load_ms, 2), "cache_dir": st.cache_dir.decode()}


#: every symbol include/ca3d.h declares: (name, restype, argtypes)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_H = C.c_void_p
SYMBOLS = [
    ("ca3d_abi_version", C.c_int, []),
    ("ca3d_last_error", C.c_char_p, []),
    ("ca3d_selftest_exception", C.c_int, [C.c_int]),
    ("ca3d_get_jit_stats", C.c_int, [C.c_void_p]),
    ("ca3d_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("ca3d_create", C.c_int, [C.c_int, C.POINTER(_H)]),
    ("ca3d_destroy", C.c_int, [_H]),
    ("ca3d_configure", C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    ("ca3d_configure_slab", C.c_int, [_H, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("ca3d_set_rules", C.c_int, [_H, _i32p, C.c_uint32, _i32p, C.c_uint32, _i32p, C.c_uint32, _u32p, _u32p]),
    ("ca3d_upload_state", C.c_int, [_H, _u32p, C.c_size_t]),
    ("ca3d_read_state", C.c_int, [_H, _u32p, C.c_size_t]),
    ("ca3d_step", C.c_int, [_H, C.c_uint32]),
    ("ca3d_flush", C.c_int, [_H]),
    ("ca3d_slab_step", C.c_int, [_H, C.c_uint32]),
    ("ca3d_slab_step_phase", C.c_int, [_H, C.c_uint32, C.c_int]),
    ("ca3d_slab_region", C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("ca3d_comm_unique_id", C.c_int, [C.c_void_p]),
    ("ca3d_slab_comm_init", C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    ("ca3d_slab_run", C.c_int, [_H, C.c_uint32, C.c_int]),
    ("ca3d_slab_exchange", C.c_int, [_H]),
    ("ca3d_slab_comm_info", C.c_int, [_H, C.c_void_p]),
    ("ca3d_slab_gather", C.c_int, [_H, _H]),
    ("ca3d_group_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_H)]),
    ("ca3d_group_destroy", C.c_int, [_H]),
    ("ca3d_group_size", C.c_int, [_H, C.POINTER(C.c_int)]),
    ("ca3d_group_engine", C.c_int, [_H, C.c_int, C.POINTER(_H)]),
    ("ca3d_group_configure", C.c_int, [_H, C.c_uint32, C.c_int, C.c_uint32]),
    ("ca3d_group_set_rules", C.c_int, [_H, _i32p, C.c_uint32, _i32p, C.c_uint32, _i32p, C.c_uint32, _u32p, _u32p]),
    ("ca3d_group_upload_state", C.c_int, [_H, _u32p, C.c_size_t]),
    ("ca3d_group_read_state", C.c_int, [_H, _u32p, C.c_size_t]),
    ("ca3d_group_step", C.c_int, [_H, C.c_uint32]),
    ("ca3d_group_synchronize", C.c_int, [_H]),
    ("ca3d_group_set_option", C.c_int, [_H, C.c_char_p, C.c_int64]),
    ("ca3d_group_render", C.c_int, [_H, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ca3d_render_target", C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("ca3d_synchronize", C.c_int, [_H]),
    ("ca3d_recovered_launches", C.c_int, [_H, C.POINTER(C.c_uint32)]),
    ("ca3d_measure_copy", C.c_int, [_H, C.c_size_t, C.c_uint32, C.POINTER(C.c_double)]),
    ("ca3d_set_stream", C.c_int, [_H, C.c_void_p]),
    ("ca3d_use_own_stream", C.c_int, [_H]),
    ("ca3d_device_buffer", C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("ca3d_get_info", C.c_int, [_H, C.POINTER(Info)]),
    ("ca3d_get_jit_log", C.c_int, [_H, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ca3d_get_kernel_variant", C.c_int, [_H, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ca3d_get_stats", C.c_int, [_H, C.POINTER(Stats)]),
    ("ca3d_set_option", C.c_int, [_H, C.c_char_p, C.c_int64]),
    ("ca3d_render", C.c_int, [_H, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ca3d_get_render_pipeline", C.c_int, [_H, C.POINTER(C.c_int32)]),
    ("ca3d_get_render_stats", C.c_int, [_H, C.POINTER(RenderStats)]),
]

_lib = None


def _pin_hip_runtime() -> None:
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's); libca3d.so needs `libamdhip64.so.7` by SONAME. If torch is
    imported first the dynamic linker hands libca3d.so torch's copy and everything shares one runtime; in the
    opposite order torch would load a second runtime and find no GPU. So when torch is installed, import it
    before the library (tests, bench.py and torch.distributed need it in the same process anyway)."""
    if os.environ.get("CA3D_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load() -> C.CDLL:
    """Load libca3d.so (built by `__graft_entry__.build()` / `make -C cellularautomatons3d_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
                "cellularautomatons3d_amd has no CPU fallback.")
        _pin_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != CA3D_OK:
        raise Ca3dError(rc, load().ca3d_last_error().decode("utf-8", "replace"))
