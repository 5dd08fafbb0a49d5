"""ctypes wrapper over oracle/libca3d_oracle.so — TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libca3d_oracle.so")
_lib = None

u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = C.CDLL(_SO)
        _lib.ca3d_oracle_fnv1a32.restype = C.c_uint32
        _lib.ca3d_oracle_fnv1a32.argtypes = [C.c_void_p, C.c_size_t]
        _lib.ca3d_oracle_popcount.restype = C.c_uint64
        _lib.ca3d_oracle_popcount.argtypes = [C.c_void_p, C.c_size_t]
        _lib.ca3d_oracle_fill.restype = None
        _lib.ca3d_oracle_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    return _lib


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(u32p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


class Rules:
    """The five arrays the reference binds as group 2 (main_pathtraced.js:1647-1673)."""

    def __init__(self, main, edges, corners, survive, born):
        self.main = np.ascontiguousarray(main, dtype=np.int32)
        self.edges = np.ascontiguousarray(edges, dtype=np.int32)
        self.corners = np.ascontiguousarray(corners, dtype=np.int32)
        self.survive = np.ascontiguousarray(survive, dtype=np.uint32)
        self.born = np.ascontiguousarray(born, dtype=np.uint32)

    @staticmethod
    def from_strings(neighbourhood="von neumann", born="1,3", survive="0-6", born_edges="27", survive_edges="27",
                     born_corners="27", survive_corners="27"):
        from cellularautomatons3d_amd import host

        b, s = host.recalculate_rules_values(born, survive, born_edges, survive_edges, born_corners, survive_corners)
        return Rules(host.NEIGHBOURHOOD_MAP[neighbourhood], host.NEIGHBOURHOOD_MAP["edges"],
                     host.NEIGHBOURHOOD_MAP["corners"], s, b)

    def cargs(self):
        return (self.main.ctypes.data_as(i32p), C.c_uint32(self.main.size),
                self.edges.ctypes.data_as(i32p), C.c_uint32(self.edges.size),
                self.corners.ctypes.data_as(i32p), C.c_uint32(self.corners.size),
                self.survive.ctypes.data_as(u32p), self.born.ctypes.data_as(u32p))


def packed_step_literal(G, state, rules):
    a, ap = _u32(state)
    out = np.empty_like(a)
    rc = lib().ca3d_oracle_packed_step_literal(C.c_uint32(G), ap, out.ctypes.data_as(u32p), *rules.cargs())
    assert rc == 0, rc
    return out


def packed_step(G, state, rules, nthreads=0):
    a, ap = _u32(state)
    out = np.empty_like(a)
    rc = lib().ca3d_oracle_packed_step_fast(C.c_uint32(G), ap, out.ctypes.data_as(u32p), *rules.cargs(), C.c_int(nthreads))
    assert rc == 0, rc
    return out


def packed_run(G, state, rules, steps, nthreads=0):
    """`steps` successive steps from `state`; returns the final state."""
    cur = np.ascontiguousarray(state, dtype=np.uint32).copy()
    nxt = np.empty_like(cur)
    for _ in range(steps):
        rc = lib().ca3d_oracle_packed_step_fast(C.c_uint32(G), cur.ctypes.data_as(u32p), nxt.ctypes.data_as(u32p),
                                                 *rules.cargs(), C.c_int(nthreads))
        assert rc == 0, rc
        cur, nxt = nxt, cur
    return cur


def packed_step_planes(G, planes, zbase, lo, hi, rules, wrap_full=False, nthreads=0):
    """Step output planes [lo, hi) of a plane array with ghosts; other planes of the result are zero."""
    a, ap = _u32(planes)
    words_per_plane = (G // 32) * G
    assert a.size % words_per_plane == 0
    nplanes = a.size // words_per_plane
    out = np.zeros_like(a)
    rc = lib().ca3d_oracle_packed_step_planes(C.c_uint32(G), ap, out.ctypes.data_as(u32p), C.c_int64(zbase),
                                              C.c_uint32(nplanes), C.c_uint32(lo), C.c_uint32(hi),
                                              C.c_int(1 if wrap_full else 0), *rules.cargs(), C.c_int(nthreads))
    assert rc == 0, rc
    return out


def unpacked_step(G, state, offs, survive, born, nthreads=0):
    a, ap = _u32(state)
    o, op = _i32(offs)
    s, sp = _u32(survive)
    b, bp = _u32(born)
    out = np.empty_like(a)
    rc = lib().ca3d_oracle_unpacked_step(C.c_uint32(G), ap, out.ctypes.data_as(u32p), op, C.c_uint32(o.size),
                                         sp, C.c_uint32(s.size), bp, C.c_uint32(b.size), C.c_int(nthreads))
    assert rc == 0, rc
    return out


def unpacked_step_planes(G, planes, lo, hi, offs, survive, born):
    a, ap = _u32(planes)
    o, op = _i32(offs)
    s, sp = _u32(survive)
    b, bp = _u32(born)
    nplanes = a.size // (G * G)
    out = np.zeros_like(a)
    rc = lib().ca3d_oracle_unpacked_step_planes(C.c_uint32(G), ap, out.ctypes.data_as(u32p), C.c_uint32(nplanes),
                                                C.c_uint32(lo), C.c_uint32(hi), op, C.c_uint32(o.size),
                                                sp, C.c_uint32(s.size), bp, C.c_uint32(b.size))
    assert rc == 0, rc
    return out


def fnv1a32(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().ca3d_oracle_fnv1a32(a.ctypes.data, a.nbytes))


def popcount(arr):
    a = np.ascontiguousarray(arr, dtype=np.uint32)
    return int(lib().ca3d_oracle_popcount(a.ctypes.data, a.size))


def fill(n_words, seed=0xCA3D0001, and_rounds=0):
    out = np.empty(n_words, dtype=np.uint32)
    lib().ca3d_oracle_fill(out.ctypes.data, n_words, seed, and_rounds)
    return out


def render(cells, G, uniforms, W, H, spp=1, rows=None, legacy=False, indirect=False):
    """Oracle frame: (light f32[H,W,4], depth f32[H,W,2], presentation f32[H,W,4], shadow_rays)."""
    c, cp = _u32(cells)
    u = np.ascontiguousarray(uniforms, dtype=np.float32)
    light = np.zeros((H, W, 4), dtype=np.float32)
    depth = np.zeros((H, W, 2), dtype=np.float32)
    pres = np.zeros((H, W, 4), dtype=np.float32)
    y0, y1 = rows if rows else (0, H)
    fn = lib().ca3d_oracle_render_legacy if legacy else (lib().ca3d_oracle_render_indirect if indirect else lib().ca3d_oracle_render)
    fn.restype = C.c_int64
    n = fn(cp, C.c_uint32(G), u.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(W), C.c_uint32(H), C.c_uint32(spp),
           light.ctypes.data_as(C.POINTER(C.c_float)), depth.ctypes.data_as(C.POINTER(C.c_float)),
           pres.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(y0), C.c_uint32(y1))
    assert n >= 0, n
    return light, depth, pres, int(n)


def primary_bruteforce(cells, G, uniforms, W, H, px, py):
    c, cp = _u32(cells)
    u = np.ascontiguousarray(uniforms, dtype=np.float32)
    cell = C.c_int64(-1)
    fn = lib().ca3d_oracle_primary_bruteforce
    fn.restype = C.c_float
    d = fn(cp, C.c_uint32(G), u.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(W), C.c_uint32(H), C.c_uint32(px),
           C.c_uint32(py), C.byref(cell))
    return float(d), int(cell.value)


BRANCH_VOLUME, BRANCH_DEPTH_REPAIR, BRANCH_UV_OUTSIDE, BRANCH_CELL_DIFFERS, BRANCH_BLENDED, BRANCH_LIT = 1, 2, 4, 8, 16, 32


def render_frame(cells, G, uniforms, W, H, prev_light=None, prev_depth=None, branches=False):
    """One literal reference frame (stochastic march + temporal history). prev_* are what the previous frame wrote,
    already rounded to binary16 (float arrays [H,W,4] / [H,W,2]) or None for an empty history. branches=True appends a
    uint8 [H,W] array of BRANCH_* flags: which branches of R6 / R10 each pixel took."""
    c, cp = _u32(cells)
    u = np.ascontiguousarray(uniforms, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    light = np.zeros((H, W, 4), dtype=np.float32)
    depth = np.zeros((H, W, 2), dtype=np.float32)
    pres = np.zeros((H, W, 4), dtype=np.float32)
    pl = np.ascontiguousarray(prev_light, dtype=np.float32) if prev_light is not None else None
    pd = np.ascontiguousarray(prev_depth, dtype=np.float32) if prev_depth is not None else None
    br = np.zeros((H, W), dtype=np.uint8) if branches else None
    rc = lib().ca3d_oracle_render_frame_branches(cp, C.c_uint32(G), u.ctypes.data_as(fp), C.c_uint32(W), C.c_uint32(H),
                                                 pl.ctypes.data_as(fp) if pl is not None else None,
                                                 pd.ctypes.data_as(fp) if pd is not None else None,
                                                 light.ctypes.data_as(fp), depth.ctypes.data_as(fp), pres.ctypes.data_as(fp),
                                                 br.ctypes.data_as(C.POINTER(C.c_uint8)) if branches else None)
    assert rc == 0, rc
    return (light, depth, pres, br) if branches else (light, depth, pres)
