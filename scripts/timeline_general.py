#!/usr/bin/env python3
"""Phase timeline of the general reverb path (second half of the OALSFX_DEBUG_TIMELINE file).
usage: python scripts/timeline_general.py <file> [clock_MHz]"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
g = raw[64 * 4 * 96:].reshape(64, 96)
mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
names = ["tile start->requests", "send mix", "shelves", "early", "late", "outputs (to next tile start)"]
rows = []
for w in g:
    n = int(np.count_nonzero(w))
    if n < 8:
        continue
    ts = w[:n].astype(np.int64)
    tiles = (n - 2) // 6
    pro = ts[1] - ts[0]
    seg = np.zeros(6)
    for t in range(tiles):
        s = ts[2 + 6 * t: 2 + 6 * (t + 1)]
        nxt = ts[2 + 6 * (t + 1)] if 2 + 6 * (t + 1) < n else s[5]
        seg += np.array([s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], nxt - s[5]])
    rows.append((pro, seg / max(tiles, 1), tiles, ts[n - 1] - ts[0]))
print(f"sampled general-path instances: {len(rows)}  (clock {mhz} MHz)")
if rows:
    print(f"prologue {np.mean([r[0] for r in rows]) / mhz:.2f} us, first to last stamp {np.mean([r[3] for r in rows]) / mhz:.2f} us, tiles {np.mean([r[2] for r in rows]):.1f}")
    seg = np.mean([r[1] for r in rows], axis=0) / mhz
    for nme, v in zip(names, seg):
        print(f"  {nme:32s} {v:7.2f} us per tile")
    print(f"  sum {seg.sum():.2f} us per tile")
