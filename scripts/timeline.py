#!/usr/bin/env python3
"""Phase timeline of k_reverb_steady_coop from the stamps a run with OALSFX_DEBUG_TIMELINE=<file> leaves behind.

usage: python scripts/timeline.py <file> [clock_MHz]
Per tile of an EAX instance: 8 barriers, a stamp before and after each, one at the end of the tile.  Work segments are
P1, C1, P2, C2, P3, C3, P4, C4, P5 (C* are the chain phases: real work only on the duty wave); "bar" is the wait at the
barrier that ends the segment."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)[:64 * 4 * 96].reshape(-1, 4, 96)  # the rest of the file belongs to timeline_general.py
mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
names = ["P1", "C1", "P2", "C2", "P3", "C3", "P4", "C4", "P5"]
work = {n: [] for n in names}
wait = {n: [] for n in names[:-1]}
tile_total, kernel_total, prologue, epilogue = [], [], [], []
for wg in raw:
    if wg[0, 0] == 0:
        continue
    for w in wg:
        n = int(np.count_nonzero(w))
        ts = w[:n].astype(np.int64)
        per_tile = 17
        head = 5  # start, descriptors read, tables written, barrier, first inputs arrived
        tiles = (n - head - 1) // per_tile
        kernel_total.append(ts[n - 1] - ts[0])
        prologue.append(ts[1:head] - ts[0:head - 1])
        epilogue.append(ts[n - 1] - ts[n - 2])
        prev = ts[head - 1]
        for t in range(tiles):
            s = ts[head + t * per_tile: head + (t + 1) * per_tile]
            start = prev
            for k in range(8):
                work[names[k]].append(s[2 * k] - prev)
                wait[names[k]].append(s[2 * k + 1] - s[2 * k])
                prev = s[2 * k + 1]
            work["P5"].append(s[16] - prev)
            prev = s[16]
            if t > 0:
                tile_total.append(prev - start)
us = lambda v: float(np.mean(v)) / mhz
print(f"sampled waves: {len(kernel_total)}   clock {mhz} MHz")
pro = np.mean(np.array(prologue), axis=0) / mhz
print(f"kernel (first to last stamp): {us(kernel_total):8.2f} us   tile (tiles 1..): {us(tile_total):7.2f} us")
print(f"prologue: descriptors+test {pro[0]:.2f}, tables {pro[1]:.2f}, first requests {pro[2]:.2f}, barrier {pro[3]:.2f} us; epilogue {us(epilogue):.2f} us")
tw = tb = 0.0
for n in names:
    w = us(work[n]); b = us(wait[n]) if n in wait else 0.0
    tw += w; tb += b
    print(f"  {n}: work {w:6.2f} us   barrier wait {b:6.2f} us   (max work {np.max(work[n]) / mhz:6.2f})")
print(f"  sum: work {tw:6.2f} us, barrier wait {tb:6.2f} us per tile")
