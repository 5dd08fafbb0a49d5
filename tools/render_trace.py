#!/usr/bin/env python3
"""Reads the per-wave trace the renderer writes under CA3D_RENDER_TRACE=<file> ({start, end} in s_memrealtime ticks of
10 ns, HW_ID, cell visits per wave of the scheduled kernel) and prints how the frame filled the chip: waves resident over
time, wave lifetimes, work per wave, waves per CU."""
import sys

import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64)[8:].reshape(-1, 4)
t = t[t[:, 1] > 0]
start, end, hw, vis = (t[:, i].astype(np.int64) for i in range(4))
t0 = start.min()
start, end = (start - t0) * 10e-3, (end - t0) * 10e-3  # us
life = end - start
print(f"{len(t)} waves, frame {end.max():.0f} us; wave lifetime: mean {life.mean():.1f} us, median {np.median(life):.1f}, p90 {np.percentile(life, 90):.1f}, "
      f"max {life.max():.1f}; cell visits per wave: mean {vis.mean():.0f}, max {vis.max()}")
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se in [15:13], xcc via XCC_ID reg)
xcc = (hw >> 32) & 0xF
cuid = ((hw >> 8) & 0xFF) | (xcc << 8)  # cu, sh, se bits + XCC: one id per CU
ids = np.unique(cuid)
print("distinct CUs seen:", len(ids))
# concurrency per CU at sample times
samples = np.linspace(0, end.max(), 200)
per_cu_max = np.zeros(len(ids), dtype=np.int64)
chip = []
for s_ in samples:
    live = (start <= s_) & (end > s_)
    c = np.bincount(np.searchsorted(ids, cuid[live]), minlength=len(ids))
    per_cu_max = np.maximum(per_cu_max, c)
    chip.append((int(c.max()), int(c.min()), float(c.mean())))
print("waves per CU, max over time and CUs:", int(per_cu_max.max()), "; per-CU maxima: min", int(per_cu_max.min()), "mean", float(per_cu_max.mean()))
for i in range(0, 200, 10):
    print(f"  t={samples[i]:6.0f} us: per-CU waves max {chip[i][0]} min {chip[i][1]} mean {chip[i][2]:.1f}")
work_cu = np.bincount(np.searchsorted(ids, cuid), weights=vis.astype(np.float64), minlength=len(ids))
print(f"cell visits per CU: min {work_cu.min():.0f} mean {work_cu.mean():.0f} max {work_cu.max():.0f}")
last_end = np.array([end[cuid == i].max() for i in ids])
print(f"last wave end per CU: min {last_end.min():.0f} mean {last_end.mean():.0f} max {last_end.max():.0f} us")
edges = np.linspace(0, end.max(), 41)
print("time us : waves resident (chip-wide), started, mean visits of waves started")
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = int(((start <= mid) & (end > mid)).sum())
    st = (start >= a) & (start < b)
    print(f"{a:7.0f} : {res:5d} {int(st.sum()):6d} {vis[st].mean() if st.any() else 0:8.0f}")
order = np.argsort(-life)[:10]
print("longest waves (index, start, life, visits):", [(int(i), round(float(start[i])), round(float(life[i])), int(vis[i])) for i in order])
busy = life.sum()
print(f"sum of lifetimes {busy / 1e3:.1f} ms -> mean residency {busy / end.max():.0f} waves of 6144 slots (24 per CU)")
