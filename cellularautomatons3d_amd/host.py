"""Host-side surface of the CA path: rule strings, neighbourhood tables, packed state layout, seeds.

Python mirror of the reference host logic (same names, argument meaning and quirks), used by the tests and
`bench.py` above the C ABI. The JavaScript twin is `js/ca3d.js`. All citations are to
/root/reference/main_pathtraced.js.

Nothing here touches the GPU; nothing here is a CPU fallback for the kernels.
"""
from __future__ import annotations

import math
import re
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

NEIGHBOURS_STORAGE_LEN = 27  # main_pathtraced.js:10
WORK_GROUP_SIZE = 16  # main_pathtraced.js:5

# Offset tables, flat xyz triples (main_pathtraced.js:13-85). Element order is kept: the reference's kernel
# sums over them so order is irrelevant to the result, but the ABI passes them verbatim.
_VN = [1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1]
_VN2D = [1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0]
_MOORE2D = [1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 1, 1, 0, -1, 1, 0, 1, -1, 0, -1, -1, 0]
_MOORE = (
    _MOORE2D
    + [1, 0, 1, -1, 0, 1, 0, 1, 1, 0, -1, 1, 1, 1, 1, -1, 1, 1, 1, -1, 1, -1, -1, 1, 0, 0, 1]
    + [1, 0, -1, -1, 0, -1, 0, 1, -1, 0, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 0, 0, -1]
)
_EDGES = [1, 1, 0, -1, 1, 0, 0, 1, 1, 0, 1, -1, 1, -1, 0, -1, -1, 0, 0, -1, 1, 0, -1, -1, 1, 0, 1, -1, 0, 1, 1, 0, -1, -1, 0, -1]
_CORNERS = [1, 1, 1, -1, 1, 1, 1, 1, -1, -1, 1, -1, 1, -1, 1, -1, -1, 1, 1, -1, -1, -1, -1, -1]

#: main_pathtraced.js:87-94
NEIGHBOURHOOD_MAP: Dict[str, np.ndarray] = {
    "moore": np.array(_MOORE, dtype=np.int32),
    "moore 2D": np.array(_MOORE2D, dtype=np.int32),
    "von neumann": np.array(_VN, dtype=np.int32),
    "von neumann 2D": np.array(_VN2D, dtype=np.int32),
    "edges": np.array(_EDGES, dtype=np.int32),
    "corners": np.array(_CORNERS, dtype=np.int32),
}

#: Defaults of the MainModule constructor (main_pathtraced.js:101, 123-132).
DEFAULTS = {
    "gridSize": 64,
    "neighbourhood": "von neumann",
    "bornRulesString": "1,3",
    "surviveRulesString": "0-6",
    "bornRulesStringEdges": "27",
    "surviveRulesStringEdges": "27",
    "bornRulesStringCorners": "27",
    "surviveRulesStringCorners": "27",
}

_PARSE_INT = re.compile(r"^\s*([+-]?[0-9]+)")


def _js_parse_int(s: str) -> Optional[int]:
    """`parseInt(s, 10)`: leading whitespace and sign allowed, trailing junk ignored, None for NaN."""
    m = _PARSE_INT.match(s)
    return int(m.group(1)) if m else None


def rules_components_to_values(rules_components: str) -> List[int]:
    """`_rulesComponentsToValues` (main_pathtraced.js:554-581).

    Spaces are stripped, components split on ',', 'a-b' is an inclusive range, every value is clamped to 26.
    Components that `parseInt` cannot read contribute nothing (the reference pushes NaN, which its typed-array
    store then ignores).
    """
    result: List[int] = []
    components = rules_components.replace(" ", "").split(",")
    for comp in components:
        if "-" in comp:
            parts = comp.split("-")
            start = _js_parse_int(parts[0])
            end = _js_parse_int(parts[1])
            if start is None or end is None:
                continue
            if end - start > 1_000_000:
                raise ValueError("rule range too long")
            for j in range(start, end + 1):
                result.append(min(j, 26))
        else:
            v = _js_parse_int(comp)
            if v is not None:
                result.append(min(v, 26))
    return result


def recalculate_rules_values(
    born: str = DEFAULTS["bornRulesString"],
    survive: str = DEFAULTS["surviveRulesString"],
    born_edges: str = DEFAULTS["bornRulesStringEdges"],
    survive_edges: str = DEFAULTS["surviveRulesStringEdges"],
    born_corners: str = DEFAULTS["bornRulesStringCorners"],
    survive_corners: str = DEFAULTS["surviveRulesStringCorners"],
) -> Tuple[np.ndarray, np.ndarray]:
    """`_recalculateRulesValues` (main_pathtraced.js:583-622) -> (bornRulesValues, surviveRulesValues).

    Two Uint32Array(81): 27 slots per rule-set at offsets 0 / 27 / 54 (main, edges, corners).
    """
    rulesets = [born, survive, born_edges, survive_edges, born_corners, survive_corners]
    born_values = np.zeros(NEIGHBOURS_STORAGE_LEN * 3, dtype=np.uint32)
    survive_values = np.zeros(NEIGHBOURS_STORAGE_LEN * 3, dtype=np.uint32)
    offset = 0
    for i in range(0, len(rulesets), 2):
        for v in rules_components_to_values(rulesets[i]):
            if 0 <= v + offset < born_values.size:  # typed arrays drop out-of-range stores
                born_values[v + offset] = 1
        for v in rules_components_to_values(rulesets[i + 1]):
            if 0 <= v + offset < survive_values.size:
                survive_values[v + offset] = 1
        offset += NEIGHBOURS_STORAGE_LEN
    return born_values, survive_values


def grid_size_ui_formatter(v: int) -> int:
    """`_gridSizeUIFormatter` (main_pathtraced.js:675-693): round to the closest multiple of 32 (ties down)."""
    out = v
    m = v % 32
    if m > 0:
        out = v - m if m <= 16 else v - m + 32
    return out


def words_per_buffer(grid_size: int) -> int:
    """`new Uint32Array((G / 32) * G * G)` (main_pathtraced.js:1241)."""
    _check_grid(grid_size)
    return (grid_size // 32) * grid_size * grid_size


def get_cluster_idx_from_grid_coordinates(grid_size: int, x: int, y: int, z: int) -> int:
    """`_getClusterIdxFromGridCoordinates` (main_pathtraced.js:1170-1178)."""
    cols = grid_size // 32
    layer = cols * grid_size
    return ((x // 32) % cols) + (y % grid_size) * cols + (z % grid_size) * layer


def _check_grid(grid_size: int) -> None:
    if grid_size <= 0 or grid_size % 32:
        raise ValueError(f"grid size must be a positive multiple of 32, got {grid_size}")


def initial_state(
    grid_size: int, random_initial_state: bool = False, random: Optional[Callable[[], float]] = None
) -> np.ndarray:
    """Initial packed state of `_setupStorageBuffers` (main_pathtraced.js:1241-1297).

    Default: one cell at (c, c, c), c = floor(G/2) - 1. Random mode: the 5x5x5 block around c, each cell set
    iff `random() > .5`, drawn in the reference's i, j, k loop order; `random` replaces the reference's
    unseeded `Math.random` (defaults to a fixed-seed generator so runs are reproducible). The same data goes
    to both ping-pong buffers (1361-1362).
    """
    _check_grid(grid_size)
    data = np.zeros(words_per_buffer(grid_size), dtype=np.uint32)
    center = math.floor(grid_size * 0.5) - 1
    if random_initial_state:
        if random is None:
            rng = np.random.default_rng(0xCA3D0001)
            random = lambda: float(rng.random())  # noqa: E731
        for i in range(-2, 3):
            for j in range(-2, 3):
                for k in range(-2, 3):
                    idx = get_cluster_idx_from_grid_coordinates(grid_size, center + i, center + j, center + k)
                    bit = np.uint32(1 << ((center + i) & 31))  # JS `1 << center + i` masks the shift count
                    if random() > 0.5:
                        data[idx] |= bit
                    else:
                        data[idx] &= ~bit
    else:
        idx = get_cluster_idx_from_grid_coordinates(grid_size, center, center, center)
        data[idx] = np.uint32(1 << (center % 32))
    return data


def dispatch_shape(grid_size: int) -> Tuple[int, int, int]:
    """`dispatchWorkgroups(G / 32, ceil(G / 16), ceil(G / 16))` (main_pathtraced.js:1805-1806)."""
    wg = math.ceil(grid_size / WORK_GROUP_SIZE)
    return (grid_size // 32, wg, wg)


# ---------------------------------------------------------------------------------------- synthetic inputs


def _mix32(seed: int, i: np.ndarray, rnd: int) -> np.ndarray:
    x = (i.astype(np.uint64) * 0x9E3779B9 + seed + rnd * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x.astype(np.uint32)


def random_fill(n_words: int, seed: int = 0xCA3D0001, and_rounds: int = 0) -> np.ndarray:
    """Counter-based synthetic fill: word[i] = mix32(seed, i); density 2^-(1+and_rounds)."""
    i = np.arange(n_words, dtype=np.uint64)
    w = _mix32(seed, i, 0)
    for r in range(1, and_rounds + 1):
        w &= _mix32(seed, i, r)
    return w


def cells_to_words(grid_size: int, cells: Iterable[Sequence[int]]) -> np.ndarray:
    """Packed state with exactly the listed (x, y, z) cells alive."""
    data = np.zeros(words_per_buffer(grid_size), dtype=np.uint32)
    for (x, y, z) in cells:
        data[get_cluster_idx_from_grid_coordinates(grid_size, x, y, z)] |= np.uint32(1 << (x % 32))
    return data


def get_cell(grid_size: int, words: np.ndarray, x: int, y: int, z: int) -> int:
    return int((int(words[get_cluster_idx_from_grid_coordinates(grid_size, x, y, z)]) >> (x % 32)) & 1)


# ------------------------------------------------------------------------------------------------ renderer
# The 128-float common uniform block (MemoryManager.js; allocation order main_pathtraced.js:166, 467-478 ==
# struct CommonBufferLayout, pathtraced_fragment_clustered.wgsl:17-34). Matrices are column-major f32.

UNIFORM_INDEX = {
    "light": 0, "viewMat": 4, "projViewMatInv": 20, "prevViewMat": 36, "prevProjViewMatInv": 52, "windowSize": 68,
    "elapsedTime": 70, "depthSamples": 71, "shadowSamples": 72, "cellSize": 73, "showDepthOverlay": 74,
    "temporalAlpha": 75, "baseReflectivity": 76, "roughness": 79, "materialColor": 80, "gamma": 83,
}

RENDER_DEFAULTS = {  # main_pathtraced.js:116-121, 135-152, 164-165
    "light": (0.721, 1.0, 1.0, 5.0), "depthSamples": 35, "shadowSamples": 30, "cellSize": 0.85, "showDepthOverlay": 0,
    "temporalAlpha": 0.1, "baseReflectivity": (0.17, 0.17, 0.17), "roughness": 0.29, "materialColor": (0.0, 0.0, 0.0),
    "gamma": 2.0, "fov_deg": 75.0, "near": 0.01, "far": 1000.0,
}


def mat4_perspective(fov_rad: float, aspect: float, near: float, far: float) -> np.ndarray:
    """wgpu-matrix mat4.perspective (libs/wgpu-matrix.module.js:3140): depth 0..1, column-major."""
    f = np.float32(math.tan(math.pi * 0.5 - 0.5 * fov_rad))
    m = np.zeros(16, dtype=np.float32)
    m[0] = f / np.float32(aspect)
    m[5] = f
    m[11] = -1.0
    range_inv = np.float32(1.0 / (near - far))
    m[10] = np.float32(far) * range_inv
    m[14] = np.float32(far) * np.float32(near) * range_inv
    return m


def mat4_multiply(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """mat4.multiply(a, b) = a * b on column-major arrays."""
    A = np.asarray(a, dtype=np.float32).reshape(4, 4).T
    B = np.asarray(b, dtype=np.float32).reshape(4, 4).T
    return (A @ B).T.reshape(16).astype(np.float32)


def mat4_inverse(m: np.ndarray) -> np.ndarray:
    M = np.asarray(m, dtype=np.float64).reshape(4, 4).T
    return np.linalg.inv(M).T.reshape(16).astype(np.float32)


def camera_matrix(position=(0.0, 0.0, 0.75), axis=(0.0, 1.0, 0.0), angle_rad: float = 0.0) -> np.ndarray:
    """Camera-to-world `viewMat` (the shader reads cameraPos = viewMat[3].xyz, :812): rotation about `axis`
    followed by translation to `position`. Default = the reference's start pose (main_pathtraced.js:207-213)."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / np.linalg.norm(ax)
    c, s = math.cos(angle_rad), math.sin(angle_rad)
    x, y, z = ax
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = position
    return M.T.reshape(16).astype(np.float32)


def orbit_camera(distance: float = 1.4, axis=(1.0, 1.0, 0.0), angle_rad: float = 0.6) -> np.ndarray:
    """The oblique bench pose: rotate about `axis`, camera at `distance` looking at the origin."""
    m = camera_matrix((0.0, 0.0, 0.0), axis, angle_rad).reshape(4, 4).T.astype(np.float64)
    pos = m[:3, :3] @ np.array([0.0, 0.0, distance])
    return camera_matrix(tuple(pos), axis, angle_rad)


def uniform_block(width: int, height: int, view_mat: Optional[np.ndarray] = None, elapsed_time: float = 0.5,
                  prev_view_mat: Optional[np.ndarray] = None, **overrides) -> np.ndarray:
    """Fill the common block the way `_setupUniformsMemoryCPU` / `_updateMatrices` / `_updateUIValues` do
    (main_pathtraced.js:464-518, 1762-1773). `overrides` may set any key of RENDER_DEFAULTS."""
    p = dict(RENDER_DEFAULTS)
    for k, v in overrides.items():
        if k not in p:
            raise KeyError(k)
        p[k] = v
    u = np.zeros(128, dtype=np.float32)
    if view_mat is None:
        view_mat = camera_matrix()
    view_mat = np.asarray(view_mat, dtype=np.float32)
    proj = mat4_perspective(p["fov_deg"] * math.pi / 180.0, width / height, p["near"], p["far"])
    pvi = mat4_multiply(proj, mat4_inverse(view_mat))
    I = UNIFORM_INDEX
    u[I["light"]:I["light"] + 4] = p["light"]
    u[I["viewMat"]:I["viewMat"] + 16] = view_mat
    u[I["projViewMatInv"]:I["projViewMatInv"] + 16] = pvi
    if prev_view_mat is not None:
        prev_view_mat = np.asarray(prev_view_mat, dtype=np.float32)
        u[I["prevViewMat"]:I["prevViewMat"] + 16] = prev_view_mat
        u[I["prevProjViewMatInv"]:I["prevProjViewMatInv"] + 16] = mat4_multiply(proj, mat4_inverse(prev_view_mat))
    u[I["windowSize"]:I["windowSize"] + 2] = (width, height)
    u[I["elapsedTime"]] = elapsed_time
    for k in ("depthSamples", "shadowSamples", "cellSize", "showDepthOverlay", "temporalAlpha", "roughness", "gamma"):
        u[I[k]] = p[k]
    u[I["baseReflectivity"]:I["baseReflectivity"] + 3] = p["baseReflectivity"]
    u[I["materialColor"]:I["materialColor"] + 3] = p["materialColor"]
    return u


# --------------------------------------------------------------------------------------------- checkpoints
# The reference keeps its state only in GPU buffers and never reads it back; the engine's
# read_state / upload_state on the raw little-endian words IS the checkpoint payload. The file adds a header.

CHECKPOINT_MAGIC = b"CA3D"
CHECKPOINT_VERSION = 1


def save_checkpoint(path, words: np.ndarray, grid_size: int, step: int = 0, layout: int = 0) -> None:
    """magic 'CA3D' | u32 version | u32 grid size | u32 layout | u64 step | u64 word count | LE u32 words."""
    import struct

    w = np.ascontiguousarray(words, dtype="<u4")
    with open(path, "wb") as f:
        f.write(CHECKPOINT_MAGIC + struct.pack("<IIIQQ", CHECKPOINT_VERSION, grid_size, layout, step, w.size))
        f.write(w.tobytes())


def load_checkpoint(path):
    """-> (words, grid_size, step, layout); raises ValueError on a malformed file."""
    import struct

    with open(path, "rb") as f:
        head = f.read(4 + 28)
        if len(head) != 32 or head[:4] != CHECKPOINT_MAGIC:
            raise ValueError("not a CA3D checkpoint")
        version, grid_size, layout, step, n = struct.unpack("<IIIQQ", head[4:])
        if version != CHECKPOINT_VERSION:
            raise ValueError(f"unsupported checkpoint version {version}")
        expect = (grid_size // 32) * grid_size * grid_size if layout == 0 else grid_size ** 3
        if n != expect:
            raise ValueError("word count does not match the grid size")
        words = np.frombuffer(f.read(n * 4), dtype="<u4")
        if words.size != n:
            raise ValueError("truncated checkpoint")
    return words.astype(np.uint32), grid_size, step, layout
