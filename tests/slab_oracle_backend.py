"""CPU stand-in for one rank's slab, used only by the gloo tests: state lives in torch CPU tensors, the sub-steps
are computed by the oracle, the ghost exchange is the PRODUCT code (cellularautomatons3d_amd.slab)."""
import numpy as np
import torch

import oracle_lib as ol
from cellularautomatons3d_amd import LAYOUT_PACKED32, LAYOUT_UNPACKED, slab


class OracleSlab:
    def __init__(self, G, rank, world, ghost, rules, layout=LAYOUT_PACKED32, group=None, overlap=False):
        self.overlap = overlap
        self.G, self.rank, self.world, self.K, self.rules, self.layout, self.group = G, rank, world, ghost, rules, layout, group
        self.z0, self.nz = slab.slab_bounds(G, world, rank)
        self.plan = slab.halo_plan(rank, world, layout)
        self.pw = (G // 32) * G if layout == LAYOUT_PACKED32 else G * G
        self.L = self.nz + 2 * ghost
        self.buf = torch.zeros(self.L * self.pw, dtype=torch.int32)

    def upload(self, owned_words):
        self.buf.zero_()
        self.buf[self.K * self.pw:(self.K + self.nz) * self.pw] = torch.from_numpy(owned_words.view(np.int32).copy())

    def owned(self):
        return self.buf[self.K * self.pw:(self.K + self.nz) * self.pw].numpy().view(np.uint32).copy()

    def _regions(self):
        K, nz, pw = self.K, self.nz, self.pw
        return {"send_low": self.buf[K * pw:2 * K * pw], "send_high": self.buf[nz * pw:(nz + K) * pw],
                "recv_low": self.buf[0:K * pw], "recv_high": self.buf[(K + nz) * pw:(2 * K + nz) * pw]}

    def _step_range(self, src, dst, lo, hi):
        """dst[lo:hi) = one step of src (both full plane arrays), like one kernel launch of the engine."""
        if hi <= lo:
            return
        planes = src.numpy().view(np.uint32)
        if self.layout == LAYOUT_PACKED32:
            out = ol.packed_step_planes(self.G, planes, self.z0 - self.K, lo, hi, self.rules)
        else:
            out = ol.unpacked_step_planes(self.G, planes, lo, hi, self.rules.main, self.rules.survive, self.rules.born)
        dst[lo * self.pw:hi * self.pw] = torch.from_numpy(out.view(np.int32))[lo * self.pw:hi * self.pw]

    def run_overlapped(self, n_steps):
        """The product's overlapped schedule (SlabEngine.run) with the engine's two ping-pong buffers and plane
        ranges (slab.batch_ranges == ca3d_slab_step_phase): edges of the whole batch, exchange, interior."""
        dead_bottom = self.layout == LAYOUT_PACKED32 and self.z0 == 0
        slab.exchange_halos(self._regions(), self.plan, self.rank, self.group)
        bufs = [self.buf, torch.zeros_like(self.buf)]
        cur = 0
        left = n_steps
        while left > 0:
            k = min(self.K, left)
            for zone in ("low", "high"):
                for s in range(1, k + 1):
                    lo, hi = slab.batch_ranges(self.nz, self.K, k, s, dead_bottom)[zone]
                    self._step_range(bufs[(cur + s - 1) & 1], bufs[(cur + s) & 1], lo, hi)
            self.buf = bufs[(cur + k) & 1]  # regions of the buffer the batch ends in
            slab.exchange_halos(self._regions(), self.plan, self.rank, self.group)
            for s in range(1, k + 1):
                lo, hi = slab.batch_ranges(self.nz, self.K, k, s, dead_bottom)["interior"]
                self._step_range(bufs[(cur + s - 1) & 1], bufs[(cur + s) & 1], lo, hi)
            cur = (cur + k) & 1
            left -= k

    def run(self, n_steps):
        if self.overlap:
            return self.run_overlapped(n_steps)
        left = n_steps
        while left > 0:
            k = min(self.K, left)
            slab.exchange_halos(self._regions(), self.plan, self.rank, self.group)
            for s in range(1, k + 1):
                lo, hi = s, self.L - s
                if self.layout == LAYOUT_PACKED32 and self.z0 == 0:
                    lo = self.K
                planes = self.buf.numpy().view(np.uint32)
                if self.layout == LAYOUT_PACKED32:
                    out = ol.packed_step_planes(self.G, planes, self.z0 - self.K, lo, hi, self.rules)
                else:
                    out = ol.unpacked_step_planes(self.G, planes, lo, hi, self.rules.main, self.rules.survive, self.rules.born)
                self.buf = torch.from_numpy(out.view(np.int32))
            left -= k
