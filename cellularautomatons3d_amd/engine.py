"""`Engine`: the rule / grid / step surface of the reference host (main_pathtraced.js) over the C ABI.

Method names follow the reference's host methods: `restart_sim` (_restartSim, 624-637), `compute_pass`
(_computePass, 1796-1809), `setup_storage_buffers` (_setupStorageBuffers, 1228-1382).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _capi, host
from ._capi import Ca3dError, Info, RenderStats, Stats  # noqa: F401

_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)


def _as_u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _as_i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


class Engine:
    """One engine = one GPU, one HIP stream, two ping-pong state buffers."""

    def __init__(self, device: int = 0):
        self._lib = _capi.load()
        h = C.c_void_p()
        _capi.check(self._lib.ca3d_create(int(device), C.byref(h)))
        self._h = h
        self.grid_size = 0
        self.layout = _capi.LAYOUT_PACKED32

    # -- lifetime ---------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ca3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration ----------------------------------------------------------------------------------
    def configure(self, grid_size: int, layout: int = _capi.LAYOUT_PACKED32) -> None:
        _capi.check(self._lib.ca3d_configure(self._h, grid_size, grid_size, grid_size, layout))
        self.grid_size, self.layout = grid_size, layout

    def configure_slab(self, grid_size: int, z0: int, nz: int, ghost: int, layout: int = _capi.LAYOUT_PACKED32) -> None:
        _capi.check(self._lib.ca3d_configure_slab(self._h, grid_size, layout, z0, nz, ghost))
        self.grid_size, self.layout = grid_size, layout

    def set_rules(self, main_offsets, edges_offsets, corners_offsets, survive, born) -> None:
        m, e, c = _as_i32(main_offsets), _as_i32(edges_offsets), _as_i32(corners_offsets)
        s, b = _as_u32(survive), _as_u32(born)
        if s.size != _capi.LUT_LEN or b.size != _capi.LUT_LEN:
            raise ValueError("survive/born must hold 81 entries")
        _capi.check(self._lib.ca3d_set_rules(
            self._h, m.ctypes.data_as(_i32p), m.size, e.ctypes.data_as(_i32p), e.size,
            c.ctypes.data_as(_i32p), c.size, s.ctypes.data_as(_u32p), b.ctypes.data_as(_u32p)))

    def set_rule_strings(self, neighbourhood: str = host.DEFAULTS["neighbourhood"],
                         born: str = host.DEFAULTS["bornRulesString"], survive: str = host.DEFAULTS["surviveRulesString"],
                         born_edges: str = "27", survive_edges: str = "27",
                         born_corners: str = "27", survive_corners: str = "27") -> None:
        """`_recalculateRulesValues` + the rule part of `_setupStorageBuffers`."""
        b, s = host.recalculate_rules_values(born, survive, born_edges, survive_edges, born_corners, survive_corners)
        self.set_rules(host.NEIGHBOURHOOD_MAP[neighbourhood], host.NEIGHBOURHOOD_MAP["edges"],
                       host.NEIGHBOURHOOD_MAP["corners"], s, b)

    def restart_sim(self, grid_size: int, neighbourhood: str, born: str, survive: str, born_edges: str = "27",
                    survive_edges: str = "27", born_corners: str = "27", survive_corners: str = "27",
                    random_initial_state: bool = False, random=None) -> None:
        """`_restartSim` (624-637): apply parameters, reset the step counter, rebuild rules and state."""
        self.configure(grid_size)
        self.set_rule_strings(neighbourhood, born, survive, born_edges, survive_edges, born_corners, survive_corners)
        self.upload_state(host.initial_state(grid_size, random_initial_state, random))

    # -- state ----------------------------------------------------------------------------------------------
    def upload_state(self, words) -> None:
        w = _as_u32(words).ravel()
        _capi.check(self._lib.ca3d_upload_state(self._h, w.ctypes.data_as(_u32p), w.size))

    def read_state(self) -> np.ndarray:
        out = np.empty(self.info().state_words, dtype=np.uint32)
        _capi.check(self._lib.ca3d_read_state(self._h, out.ctypes.data_as(_u32p), out.size))
        return out

    def save_checkpoint(self, path) -> None:
        i = self.info()
        host.save_checkpoint(path, self.read_state(), i.grid_size, i.step, i.layout)

    def load_checkpoint(self, path) -> int:
        """Configure for the checkpoint's grid, upload its state; returns the step it was taken at (the engine's own
        counter restarts at 0, like after any upload)."""
        words, grid_size, step, layout = host.load_checkpoint(path)
        self.configure(grid_size, layout)
        self.upload_state(words)
        return step

    # -- stepping -------------------------------------------------------------------------------------------
    def step(self, n_steps: int = 1) -> None:
        _capi.check(self._lib.ca3d_step(self._h, n_steps))

    compute_pass = step

    def flush(self) -> None:
        """queue.submit (main_pathtraced.js:1850): hand the steps encoded since the last submission to the GPU. Only needed with
        `set_option("queue", n)` and a caller that records its own events on the stream — every other call on the engine
        submits first by itself."""
        _capi.check(self._lib.ca3d_flush(self._h))

    def slab_step(self, n_steps: int) -> None:
        _capi.check(self._lib.ca3d_slab_step(self._h, n_steps))

    def slab_step_phase(self, n_steps: int, phase: int) -> None:
        """`slab_step` in two phases (SLAB_PHASE_EDGES, then SLAB_PHASE_INTERIOR) so the halo exchange can run
        between them, overlapped with the interior."""
        _capi.check(self._lib.ca3d_slab_step_phase(self._h, n_steps, phase))

    def slab_region(self, region: int):
        p, n = C.c_void_p(), C.c_size_t()
        _capi.check(self._lib.ca3d_slab_region(self._h, region, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    # -- RCCL transport inside the engine (slab mode) ------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: rank 0 creates it and hands it to the other ranks (torch.distributed, a socket ...)."""
        buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
        _capi.check(_capi.load().ca3d_comm_unique_id(buf))
        return buf.raw

    def slab_comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        if len(unique_id) != _capi.COMM_ID_BYTES:
            raise ValueError("the communicator id holds 128 bytes")
        _capi.check(self._lib.ca3d_slab_comm_init(self._h, C.c_char_p(unique_id), rank, world))

    def slab_run(self, n_steps: int, overlap: bool = False) -> None:
        """n steps in batches of <= ghost sub-steps, the halo exchange (RCCL) between them — under the interior phase
        when `overlap`. Everything is enqueued by this one call."""
        _capi.check(self._lib.ca3d_slab_run(self._h, n_steps, 1 if overlap else 0))

    def slab_exchange(self) -> None:
        _capi.check(self._lib.ca3d_slab_exchange(self._h))

    def comm_info(self) -> dict:
        """Device ordinal + PCI bus id of this engine and, with an RCCL communicator (slab_comm_init), what the communicator itself
        reports: rank count, rank, device (ca3d_slab_comm_info)."""
        ci = _capi.CommInfo()
        _capi.check(self._lib.ca3d_slab_comm_info(self._h, C.byref(ci)))
        return {"device": int(ci.device), "pci_bus_id": ci.pci_bus_id.decode(), "comm_ranks": int(ci.comm_ranks),
                "comm_rank": int(ci.comm_rank), "comm_device": int(ci.comm_device)}

    def slab_gather(self, full: "Engine") -> None:
        """ncclAllGather of every rank's owned planes into `full`'s current buffer (a full-grid engine, same device)."""
        _capi.check(self._lib.ca3d_slab_gather(self._h, full._h))

    def render_target(self, which: int = 0):
        p, n = C.c_void_p(), C.c_size_t()
        _capi.check(self._lib.ca3d_render_target(self._h, which, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def device_buffer(self, which: int):
        p, n = C.c_void_p(), C.c_size_t()
        _capi.check(self._lib.ca3d_device_buffer(self._h, which, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def synchronize(self) -> None:
        _capi.check(self._lib.ca3d_synchronize(self._h))

    def recovered_launches(self) -> int:
        """Resident launches that timed out and whose steps were re-run through the per-step kernels (the state stayed valid)."""
        n = C.c_uint32()
        _capi.check(self._lib.ca3d_recovered_launches(self._h, C.byref(n)))
        return int(n.value)

    def measure_copy(self, n_bytes: int = 1 << 30, reps: int = 8) -> float:
        """GB/s (read + written) of a float4 device-to-device copy on the engine's stream: the practical HBM ceiling."""
        v = C.c_double()
        _capi.check(self._lib.ca3d_measure_copy(self._h, n_bytes, reps, C.byref(v)))
        return float(v.value)

    def set_stream(self, hip_stream: int) -> None:
        """Run on a caller-owned hipStream_t; 0 is HIP's legacy default stream."""
        _capi.check(self._lib.ca3d_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def use_own_stream(self) -> None:
        _capi.check(self._lib.ca3d_use_own_stream(self._h))

    def set_option(self, name: str, value: int) -> None:
        _capi.check(self._lib.ca3d_set_option(self._h, name.encode(), int(value)))

    # -- rendering ------------------------------------------------------------------------------------------
    def render(self, uniforms, width: int, height: int, spp: int = 1, readback: bool = True, rows=None):
        """`_renderPass` on the current state. Returns (presentation u8[H,W,4], light f16[H,W,4], depth f16[H,W,2])
        when `readback`, else None (targets stay on the device). `rows` = (begin, end): only that band of image rows
        is rendered (begin a multiple of 16) — the other rows of the returned arrays are whatever the targets held."""
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        if u.size != 128:
            raise ValueError("the common uniform block holds 128 floats")
        up = u.ctypes.data_as(C.POINTER(C.c_float))
        self.set_option("render_row_begin", rows[0] if rows else 0)
        self.set_option("render_row_end", rows[1] if rows else 0)
        if not readback:
            _capi.check(self._lib.ca3d_render(self._h, up, width, height, spp, None, None, None))
            return None
        pres = np.empty((height, width, 4), dtype=np.uint8)
        light = np.empty((height, width, 4), dtype=np.float16)
        depth = np.empty((height, width, 2), dtype=np.float16)
        _capi.check(self._lib.ca3d_render(self._h, up, width, height, spp, pres.ctypes.data, light.ctypes.data, depth.ctypes.data))
        return pres, light, depth

    def set_render_mode(self, literal_frame: bool) -> None:
        """False (default): the converged frame (exact cell walk). True: one literal reference frame per call —
        jittered fixed-step marches, history look-ups and the temporal blend of fragment_main (800-890)."""
        self.set_option("render_mode", 1 if literal_frame else 0)

    def reset_render_history(self) -> None:
        self.set_option("render_reset_history", 1)

    def render_stats(self) -> RenderStats:
        s = RenderStats()
        _capi.check(self._lib.ca3d_get_render_stats(self._h, C.byref(s)))
        return s

    def render_pipeline(self) -> int:
        """Converged frames the engine keeps in flight (option render_pipeline): 0 before the first pipelined frame or when off."""
        n = C.c_int32(0)
        _capi.check(self._lib.ca3d_get_render_pipeline(self._h, C.byref(n)))
        return int(n.value)

    def info(self) -> Info:
        i = Info()
        _capi.check(self._lib.ca3d_get_info(self._h, C.byref(i)))
        return i

    def jit_log(self) -> str:
        """Compiler log of the last failed run-time specialisation ('' when none failed): the pre-built kernels are
        then in charge and `info().kernel_name` carries no '(jit)'."""
        need = C.c_size_t()
        _capi.check(self._lib.ca3d_get_jit_log(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(1, need.value))
        _capi.check(self._lib.ca3d_get_jit_log(self._h, buf, len(buf), None))
        return buf.value.decode("utf-8", "replace")

    def kernel_variant(self) -> str:
        """Everything that decides which instruction stream the next step batch runs (kernel, grid, rule hash, resident-kernel form
        options, device-source hash): a committed profile is comparable with this run only if its string is the same."""
        need = C.c_size_t()
        _capi.check(self._lib.ca3d_get_kernel_variant(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(1, need.value))
        _capi.check(self._lib.ca3d_get_kernel_variant(self._h, buf, len(buf), None))
        return buf.value.decode("utf-8", "replace")

    def stats(self) -> Stats:
        s = Stats()
        _capi.check(self._lib.ca3d_get_stats(self._h, C.byref(s)))
        return s


class EngineGroup:
    """`ca3d_group_*`: one host thread drives the Z-slab split of a grid over several GPUs (or several slabs on one GPU) —
    the single-process form of `slab.NativeSlabEngine`, and what the JavaScript host binds (js/ca3d.js `EngineGroup`)."""

    def __init__(self, devices: Sequence[int]):
        self._lib = _capi.load()
        d = (C.c_int * len(devices))(*[int(x) for x in devices])
        h = C.c_void_p()
        _capi.check(self._lib.ca3d_group_create(d, len(devices), C.byref(h)))
        self._h = h
        self.devices = list(devices)
        self.grid_size = 0

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ca3d_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, grid_size: int, ghost: int, layout: int = _capi.LAYOUT_PACKED32) -> None:
        _capi.check(self._lib.ca3d_group_configure(self._h, grid_size, layout, ghost))
        self.grid_size = grid_size

    def set_rules(self, main_offsets, edges_offsets, corners_offsets, survive, born) -> None:
        m, e, c = _as_i32(main_offsets), _as_i32(edges_offsets), _as_i32(corners_offsets)
        s, b = _as_u32(survive), _as_u32(born)
        _capi.check(self._lib.ca3d_group_set_rules(self._h, m.ctypes.data_as(_i32p), m.size, e.ctypes.data_as(_i32p), e.size,
                                                   c.ctypes.data_as(_i32p), c.size, s.ctypes.data_as(_u32p), b.ctypes.data_as(_u32p)))

    def set_rule_strings(self, neighbourhood: str = host.DEFAULTS["neighbourhood"], born: str = host.DEFAULTS["bornRulesString"],
                         survive: str = host.DEFAULTS["surviveRulesString"], born_edges: str = "27", survive_edges: str = "27",
                         born_corners: str = "27", survive_corners: str = "27") -> None:
        b, s = host.recalculate_rules_values(born, survive, born_edges, survive_edges, born_corners, survive_corners)
        self.set_rules(host.NEIGHBOURHOOD_MAP[neighbourhood], host.NEIGHBOURHOOD_MAP["edges"], host.NEIGHBOURHOOD_MAP["corners"], s, b)

    def upload_state(self, words) -> None:
        w = _as_u32(words).ravel()
        _capi.check(self._lib.ca3d_group_upload_state(self._h, w.ctypes.data_as(_u32p), w.size))
        self._words = w.size

    def read_state(self) -> np.ndarray:
        out = np.empty(self._words, dtype=np.uint32)
        _capi.check(self._lib.ca3d_group_read_state(self._h, out.ctypes.data_as(_u32p), out.size))
        return out

    def step(self, n_steps: int = 1) -> None:
        _capi.check(self._lib.ca3d_group_step(self._h, n_steps))

    def synchronize(self) -> None:
        _capi.check(self._lib.ca3d_group_synchronize(self._h))

    def set_option(self, name: str, value: int) -> None:
        _capi.check(self._lib.ca3d_group_set_option(self._h, name.encode(), int(value)))

    def kernel_name(self, rank: int = 0) -> str:
        e = C.c_void_p()
        _capi.check(self._lib.ca3d_group_engine(self._h, rank, C.byref(e)))
        i = Info()
        _capi.check(self._lib.ca3d_get_info(e, C.byref(i)))
        return i.kernel_name.decode()

    def render(self, uniforms, width: int, height: int, spp: int = 1, readback: bool = True):
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        up = u.ctypes.data_as(C.POINTER(C.c_float))
        if not readback:
            _capi.check(self._lib.ca3d_group_render(self._h, up, width, height, spp, None, None, None))
            return None
        pres = np.empty((height, width, 4), dtype=np.uint8)
        light = np.empty((height, width, 4), dtype=np.float16)
        depth = np.empty((height, width, 2), dtype=np.float16)
        _capi.check(self._lib.ca3d_group_render(self._h, up, width, height, spp, pres.ctypes.data, light.ctypes.data, depth.ctypes.data))
        return pres, light, depth
