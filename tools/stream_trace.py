#!/usr/bin/env python3
"""Reads a CA3D_STREAM_TRACE file (render_stream.hip): per wave and walk pass {start, end, refill rounds, stepping iterations, chunks,
ticks refilling, jobs started} and prints what the two persistent walk launches were made of: how long the waves lived, how even
their shares were, how much of a wave's life went into refills, and how full the chip was over time."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
n = int(raw[0])
rec = raw[1:1 + 16 * n].reshape(2, n, 8).astype(np.float64)
for p, name in enumerate(("primary", "shadow")):
    r = rec[p]
    r = r[r[:, 1] > 0]
    t0 = r[:, 0].min()
    start, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0  # us
    life = end - start
    print(f"{name}: {len(r)} waves, launch {end.max():.1f} us; wave life mean {life.mean():.1f} p50 {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} "
          f"p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us; start skew max {start.max():.1f} us")
    print(f"   per wave: refill rounds {r[:, 2].mean():.1f}, iterations mean {r[:, 3].mean():.0f} max {r[:, 3].max():.0f}, chunks {r[:, 4].mean():.2f} (max {r[:, 4].max():.0f}), "
          f"jobs {r[:, 6].mean():.0f}, refill share of life {(r[:, 5].sum() / 100.0) / life.sum():.2f}, us per iteration {(life.sum() - r[:, 5].sum() / 100.0) / max(1.0, r[:, 3].sum()):.3f}")
    grid = np.linspace(0, end.max(), 21)
    occ = [((start <= g) & (end > g)).sum() for g in grid]
    print("   waves alive at 0..100% of the launch:", " ".join(str(o) for o in occ))
