"""Z-slab decomposition of the grid across GPUs and the ghost-plane exchange (no reference
counterpart — the reference is a single-GPU browser demo).

One process per GPU. Rank k of P owns global planes [k*G/P, (k+1)*G/P) plus `ghost` planes below and above.
A plane of the packed layout is contiguous (z is the slowest index), so a halo is one contiguous message of
`ghost * G*G/8` bytes per direction. The exchange follows the kernel's boundary semantics:

* PACKED32 (compute_clustered.wgsl:104 `<= G`): the -z face is dead, the +z face wraps to plane 0. The chain is
  open at the bottom and closed at the top — rank P-1's high ghost is rank 0's first planes, rank 0's low ghost
  is never read, and nobody sends rank P-1's last planes upward.
* UNPACKED (compute.wgsl, toroidal): a true ring in both directions.

With `ghost` = K planes the ranks exchange once per K steps and recompute the overlap (K sub-steps on a
shrinking plane range), because one RCCL send/recv round costs more than one slab step. The exchange is
overlapped with compute: a batch runs its edge zones first (`SLAB_PHASE_EDGES`: afterwards the planes the
neighbours need are final), the sends / receives are posted, and the interior — which reads no ghost plane —
runs while they are in flight (`SLAB_PHASE_INTERIOR`); the next batch waits for the receives.

The transport is torch.distributed point-to-point (backend "nccl" = RCCL over xGMI on GPU tensors, "gloo" on
CPU tensors in the tests), posted as one batch so RCCL groups the sends and receives.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from ._capi import (LAYOUT_PACKED32, LAYOUT_UNPACKED, SLAB_OWNED, SLAB_PHASE_EDGES, SLAB_PHASE_INTERIOR, SLAB_RECV_HIGH,
                    SLAB_RECV_LOW, SLAB_SEND_HIGH, SLAB_SEND_LOW)

TAG_TO_HIGH_GHOST = 0  # a rank's first owned planes travelling down to its lower neighbour's high ghost
TAG_TO_LOW_GHOST = 1   # a rank's last owned planes travelling up to its upper neighbour's low ghost


@dataclass(frozen=True)
class HaloPlan:
    send_low_to: Optional[int]     # who receives my first `ghost` owned planes (into its high ghost)
    send_high_to: Optional[int]    # who receives my last `ghost` owned planes (into its low ghost)
    recv_low_from: Optional[int]   # who fills my low ghost
    recv_high_from: Optional[int]  # who fills my high ghost


def slab_bounds(grid_size: int, world: int, rank: int) -> Tuple[int, int]:
    """(z0, nz) of rank's slab; the grid must split evenly."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if grid_size % world:
        raise ValueError(f"grid of {grid_size} planes does not split evenly over {world} ranks")
    nz = grid_size // world
    return rank * nz, nz


def halo_plan(rank: int, world: int, layout: int = LAYOUT_PACKED32) -> HaloPlan:
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    below = (rank - 1) % world
    above = (rank + 1) % world
    if layout == LAYOUT_UNPACKED:
        return HaloPlan(send_low_to=below, send_high_to=above, recv_low_from=below, recv_high_from=above)
    if layout != LAYOUT_PACKED32:
        raise ValueError("unknown layout")
    top, bottom = rank == world - 1, rank == 0
    return HaloPlan(
        send_low_to=below,                        # rank 0's first planes wrap to the top rank's high ghost
        send_high_to=None if top else above,      # z = -1 is dead: rank 0 never reads a low ghost
        recv_low_from=None if bottom else below,
        recv_high_from=above,                     # top rank: plane G wraps to plane 0 = rank 0
    )


def batch_ranges(nz: int, ghost: int, n: int, s: int, dead_bottom: bool) -> Dict[str, Tuple[int, int]]:
    """Array-plane ranges [lo, hi) that sub-step s (1..n) of an n-sub-step batch updates (array = `ghost` low ghost
    planes, nz owned planes, `ghost` high ghost planes) — the same split `ca3d_slab_step_phase` makes: 'all' is
    the whole shrinking range; 'low' / 'high' are the edge zones that end on the planes the neighbours need;
    'interior' is what lies between and grows as they shrink. `dead_bottom`: the slab owns global plane 0 of a
    packed grid, whose -z face is dead, so its low ghost is never computed. Thin slabs cannot be split: then
    'low' is 'all' and the other two are empty."""
    L, K = nz + 2 * ghost, ghost
    lo, hi = max(s, K if dead_bottom else 0), L - s
    if nz + 2 <= 2 * K + 2 * n:
        return {"all": (lo, hi), "low": (lo, hi), "high": (hi, hi), "interior": (hi, hi)}
    e_lo, e_hi = 2 * K + n - s, L - 2 * K - n + s
    return {"all": (lo, hi), "low": (lo, e_lo), "high": (e_hi, hi), "interior": (e_lo, e_hi)}


def exchange_halos(regions: Dict[str, "object"], plan: HaloPlan, rank: int, group=None, host_staging: bool = False,
                   wait: bool = True, loopback: bool = False) -> List:
    """Refresh the ghost planes. `regions` maps 'send_low', 'send_high', 'recv_low', 'recv_high' to torch tensors
    (views of the current state buffer). Sends are posted low-then-high and receives high-then-low so that the
    two messages a pair of ranks may exchange in one direction (world == 2) match in order under RCCL, which
    ignores tags; gloo uses the tags. With wait=False the posted requests are returned instead of waited for
    (RCCL: `wait()` later makes the then-current stream wait for the transfer, the host never blocks). A message
    to oneself (world == 1, or the wrap of a one-rank chain) is a device copy unless `loopback` sends it through the
    transport as well — the single-GPU rehearsal of the RCCL path."""
    import torch.distributed as dist

    if host_staging:
        # Transport without device-to-device support (gloo): bounce the planes through host memory. Test /
        # rehearsal path only; RCCL moves the device buffers directly.
        staged = {k: v.cpu() for k, v in regions.items() if k.startswith("send")}
        staged.update({k: regions[k].new_empty(regions[k].shape, device="cpu") for k in ("recv_low", "recv_high")})
        exchange_halos(staged, plan, rank, group, host_staging=False, loopback=loopback)
        if plan.recv_high_from is not None:
            regions["recv_high"].copy_(staged["recv_high"])
        if plan.recv_low_from is not None:
            regions["recv_low"].copy_(staged["recv_low"])
        return []
    ops: List = []
    if plan.send_low_to is not None:
        if plan.send_low_to == rank and not loopback:
            regions["recv_high"].copy_(regions["send_low"])
        else:
            ops.append(dist.P2POp(dist.isend, regions["send_low"], plan.send_low_to, group, TAG_TO_HIGH_GHOST))
    if plan.send_high_to is not None:
        if plan.send_high_to == rank and not loopback:
            regions["recv_low"].copy_(regions["send_high"])
        else:
            ops.append(dist.P2POp(dist.isend, regions["send_high"], plan.send_high_to, group, TAG_TO_LOW_GHOST))
    if plan.recv_high_from is not None and (plan.recv_high_from != rank or loopback):
        ops.append(dist.P2POp(dist.irecv, regions["recv_high"], plan.recv_high_from, group, TAG_TO_HIGH_GHOST))
    if plan.recv_low_from is not None and (plan.recv_low_from != rank or loopback):
        ops.append(dist.P2POp(dist.irecv, regions["recv_low"], plan.recv_low_from, group, TAG_TO_LOW_GHOST))
    if not ops:
        return []
    reqs = dist.batch_isend_irecv(ops)
    if wait:
        for req in reqs:
            req.wait()
        return []
    return reqs


class _DevicePtr:
    """Exposes engine-owned device memory to torch through the CUDA array interface (no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def device_tensor(ptr: int, nbytes: int, device: int):
    import torch

    return torch.as_tensor(_DevicePtr(ptr, nbytes), device=f"cuda:{device}")


class SlabEngine:
    """One rank's slab on one GPU: an `Engine` in slab mode plus the halo exchange.

    Engine kernels and RCCL operations are ordered on one dedicated HIP stream: the engine is bound to it
    (`ca3d_set_stream`) and the torch.distributed calls are issued with it as torch's current stream, so torch's
    own event bookkeeping serialises "exchange -> K sub-steps -> exchange". A K-step batch is one hipGraph launch
    (`ca3d_slab_step`), so the host cost per batch is one graph launch plus one grouped send/recv.
    """

    def __init__(self, grid_size: int, rank: int, world: int, ghost: int, layout: int = LAYOUT_PACKED32,
                 device: int = 0, group=None, engine=None, host_staging: bool = False, overlap="auto",
                 loopback: bool = False):
        import torch

        from .engine import Engine

        self.rank, self.world, self.ghost, self.layout, self.group = rank, world, ghost, layout, group
        self.grid_size = grid_size
        self.z0, self.nz = slab_bounds(grid_size, world, rank)
        self.plan = halo_plan(rank, world, layout)
        self.device = device
        self.host_staging = host_staging
        torch.cuda.set_device(device)
        self.stream = torch.cuda.Stream(device=device)
        self.engine = engine or Engine(device)
        self.engine.configure_slab(grid_size, self.z0, self.nz, ghost, layout)
        self.engine.set_stream(self.stream.cuda_stream)
        # Overlapping the exchange costs two smaller launches per sub-step instead of one. Measured on one MI355X
        # with the exchange through RCCL (tools/run_slab_rccl.py): it loses on slabs of 1024^2 planes (128 owned
        # planes: 12.9 vs 10.9 us per step; 512 planes: 26.4 vs 24.7) and wins once a slab is a 1024^3-sized
        # problem (256 planes of 2048^2: 47.7 vs 60.8 us per step). "auto" draws the line at 96 MiB of owned state.
        if overlap == "auto":
            info = self.engine.info()
            overlap = info.state_words * 4 >= (96 << 20)
        self.overlap = bool(overlap)
        self.loopback = loopback
        self._regions = [None, None]
        self._parity = 0  # buffer `slab_region` refers to, mirrored here to save a round trip per batch
        self._ghosts_valid = False

    def upload_state(self, owned_words) -> None:
        self.engine.upload_state(owned_words)
        self._parity = 0
        self._ghosts_valid = False

    def _current_regions(self):
        if self._regions[self._parity] is None:
            names = {"send_low": SLAB_SEND_LOW, "send_high": SLAB_SEND_HIGH, "recv_low": SLAB_RECV_LOW,
                     "recv_high": SLAB_RECV_HIGH, "owned": SLAB_OWNED}
            self._regions[self._parity] = {k: device_tensor(*self.engine.slab_region(v), self.device) for k, v in names.items()}
        return self._regions[self._parity]

    def exchange(self, wait: bool = True) -> List:
        import torch

        with torch.cuda.stream(self.stream):
            return exchange_halos(self._current_regions(), self.plan, self.rank, self.group, self.host_staging, wait=wait,
                                  loopback=self.loopback)

    def run(self, n_steps: int) -> None:
        """n CA steps in batches of up to `ghost` sub-steps. Asynchronous on the GPU. Per batch: edge zones ->
        post the exchange of the fresh edge planes -> interior (overlaps the transfer) -> the stream waits for the
        receives. Ghosts are valid for the current step on entry (first call: one blocking exchange) and on exit."""
        import torch

        if not self._ghosts_valid:
            self.exchange()
            self._ghosts_valid = True
        left = n_steps
        while left > 0:
            k = min(self.ghost, left)
            if self.overlap:
                self.engine.slab_step_phase(k, SLAB_PHASE_EDGES)
                self._parity = (self._parity + k) & 1  # slab_region now refers to the buffer the batch ends in
                reqs = self.exchange(wait=False)
                self.engine.slab_step_phase(k, SLAB_PHASE_INTERIOR)
                with torch.cuda.stream(self.stream):
                    for req in reqs:
                        req.wait()
            else:
                self.engine.slab_step(k)
                self._parity = (self._parity + k) & 1
                self.exchange()
            left -= k

    def close(self) -> None:
        self.engine.close()


class NativeSlabEngine(SlabEngine):
    """`SlabEngine` with the halo exchange inside the engine: libca3d.so drives RCCL itself (ncclSend / ncclRecv grouped
    on the engine's stream, or on its second stream under the interior phase), so a whole `run(n)` is ONE call into the
    library — no Python, no torch.distributed bookkeeping per batch. torch.distributed only carries the 128-byte
    communicator id from rank 0 to the others (world > 1)."""

    native = True

    def __init__(self, grid_size: int, rank: int, world: int, ghost: int, layout: int = LAYOUT_PACKED32, device: int = 0,
                 group=None, overlap="auto", unique_id: Optional[bytes] = None):
        # With the exchange issued by the engine one round costs ~45 us on the GPU and ~20 us of host time, 3-18 % of a batch:
        # splitting every sub-step into an edge and an interior launch costs more than hiding that recovers at every size
        # measured (profiles/r2_a_slab_rccl_loopback.txt: 7.7 vs 10.5 us per step on 128 planes of 1024^2, 83 vs 88 on 256
        # planes of 2048^2), so "auto" means unsplit here; overlap=True remains available (BASELINE configs[4]).
        if overlap == "auto":
            overlap = False
        super().__init__(grid_size, rank, world, ghost, layout=layout, device=device, group=group, overlap=overlap)
        from .engine import Engine

        if unique_id is None:
            unique_id = Engine.comm_unique_id() if rank == 0 else b""
            if world > 1:
                import torch.distributed as dist

                box = [unique_id]
                dist.broadcast_object_list(box, src=0, group=group)
                unique_id = box[0]
        self.engine.slab_comm_init(unique_id, rank, world)

    def upload_state(self, owned_words) -> None:
        self.engine.upload_state(owned_words)

    def run(self, n_steps: int) -> None:
        self.engine.slab_run(n_steps, self.overlap)

    def exchange(self, wait: bool = True):
        self.engine.slab_exchange()
        return []


def band_rows(height: int, world: int, rank: int) -> Tuple[int, int]:
    """Image rows [begin, end) rank renders when `world` GPUs share a frame: bands of whole 16-row tiles, as even as
    the tile count allows (ranks beyond the tile count get an empty band)."""
    tiles = (height + 15) // 16
    lo, hi = (tiles * rank) // world, (tiles * (rank + 1)) // world
    return min(lo * 16, height), min(hi * 16, height)


class SlabRenderer:
    """Rendering a Z-slabbed grid: shadow rays cross slabs, so the renderer does not shard by slab.
    Every rank all-gathers the packed volume (128 MiB at 1024^3, 1 GiB at 2048^3 — RCCL all-gather straight between
    the engines' device buffers), renders its band of image rows from the full volume ("replicas over pixels") and
    the bands are gathered on rank 0. The band frame is bit-identical to the same rows of a single-GPU frame."""

    def __init__(self, slab_engine: "SlabEngine", group=None):
        import torch

        from .engine import Engine

        self.se = slab_engine
        self.group = group
        self.device = slab_engine.device
        if slab_engine.layout != LAYOUT_PACKED32:
            raise ValueError("the multi-GPU renderer takes the packed layout")
        self.full = Engine(self.device)
        self.full.configure(slab_engine.grid_size)
        self.full.set_stream(slab_engine.stream.cuda_stream)
        G = slab_engine.grid_size
        # a full-grid engine needs rules and a state before it renders; the state is overwritten by the gather
        self.full.set_rule_strings()
        self.full.upload_state(np.zeros((G // 32) * G * G, dtype=np.uint32))
        self._vol = device_tensor(*self.full.device_buffer(0), self.device)
        self._torch = torch

    def gather_volume(self) -> None:
        """All ranks: owned planes of the current state -> the full volume, in z order (= rank order)."""
        import torch.distributed as dist

        torch = self._torch
        if getattr(self.se, "native", False):
            self.se.engine.slab_gather(self.full)  # ncclAllGather inside the engine, on its stream
            return
        owned = self.se._current_regions()["owned"]
        with torch.cuda.stream(self.se.stream):
            if self.se.world == 1 and not self.se.loopback:
                self._vol.copy_(owned)
            elif self.se.host_staging:
                parts = [torch.empty(owned.shape, dtype=owned.dtype) for _ in range(self.se.world)]
                dist.all_gather(parts, owned.cpu(), group=self.group)
                self._vol.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(self._vol, owned, group=self.group)

    def render(self, uniforms, width: int, height: int, spp: int = 1):
        """Gather the volume, render this rank's band, gather the bands: rank 0 gets the presentation frame
        (u8[H, W, 4]), the other ranks None."""
        import torch.distributed as dist

        torch = self._torch
        self.gather_volume()
        world, rank = self.se.world, self.se.rank
        y0, y1 = band_rows(height, world, rank)
        if world == 1 or self.se.host_staging:
            # one rank, or the rehearsal transport (gloo): bands go through host memory
            band = np.zeros((0, width, 4), dtype=np.uint8)
            if y1 > y0:
                pres, _, _ = self.full.render(uniforms, width, height, spp, rows=(y0, y1))
                band = np.ascontiguousarray(pres[y0:y1])
            if world == 1:
                return band
            bands = [None] * world if rank == 0 else None
            dist.gather_object(band, bands, dst=0, group=self.group)
            return np.concatenate(bands, axis=0) if rank == 0 else None
        # RCCL: the bands travel between the GPUs as RGBA8 rows (a gather of equal windows: every rank sends the same
        # number of rows, starting at its band, clamped so that the window stays inside the target), no pickled host objects
        rows = max(band_rows(height, world, k)[1] - band_rows(height, world, k)[0] for k in range(world))
        start = min(y0, height - rows)
        # Everything below runs on the engines' stream: the render was enqueued there, so the window copy, the tensors the
        # gather fills and the read-back must be ordered behind it (on torch's default stream they would race the kernel and
        # the collective); the collective's own stream is ordered against this one by torch.distributed.
        with torch.cuda.stream(self.se.stream):
            if y1 > y0:
                self.full.render(uniforms, width, height, spp, readback=False, rows=(y0, y1))
                ptr, nbytes = self.full.render_target(0)
                target = device_tensor(ptr, nbytes, self.device).view(torch.uint8).view(height, width * 4)
                mine = target[start:start + rows].contiguous()
            else:
                mine = torch.zeros((rows, width * 4), dtype=torch.uint8, device=f"cuda:{self.device}")
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, parts, dst=0, group=self.group)
            if rank != 0:
                return None
            host_parts = [p.to("cpu", non_blocking=False) for p in parts]  # a blocking copy on this stream: behind the gather
        self.se.stream.synchronize()
        frame = np.empty((height, width, 4), dtype=np.uint8)
        for k in range(world):
            b0, b1 = band_rows(height, world, k)
            s0 = min(b0, height - rows)
            if b1 > b0:
                frame[b0:b1] = host_parts[k].numpy().reshape(rows, width, 4)[b0 - s0:b1 - s0]
        return frame

    def close(self) -> None:
        self.full.close()
