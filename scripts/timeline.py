#!/usr/bin/env python3
"""Phase timeline of k_reverb_steady_coop from the stamps a run with OALSFX_DEBUG_TIMELINE=<file> leaves behind.

usage: python scripts/timeline.py <file> [kernel_us]
The tile loop is skewed: iteration i runs the input half of tile i beside the late half of tile i - 1 in five steps,
S1 (P3 late + P1 input) | S2 (chain phases C1, C3 on two wavefronts) | S3 (P2, P4) | S4 (C2, C4) | S5 (P5), a workgroup barrier
after each of the first four.  Per wavefront: 5 stamps before the loop (start, record / descriptors read and tables written, -,
first requests issued, first barrier passed), 9 per iteration (end of each step's work and the barrier behind it), 1 at the end.
The shader clock's rate differs from box to box: give the kernel's duration in microseconds (bench.py's kernel_us of the same
run) to have the figures in microseconds, otherwise they are in clock ticks."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)[:64 * 4 * 96].reshape(-1, 4, 96)  # the rest of the file belongs to timeline_general.py
kernel_us = float(sys.argv[2]) if len(sys.argv) > 2 else None
names = ["S1", "S2", "S3", "S4", "S5"]
work = {n: [] for n in names}
wait = {n: [] for n in names[:-1]}
it_total, kernel_total, prologue, epilogue, first_it, last_it = [], [], [], [], [], []
duty = {"S2": [], "S4": []}   # the two longest of a workgroup's four wavefronts in a chain step: the ones that ran the recurrences
for wg in raw:
    if wg[0, 0] == 0:
        continue
    per_wave = {"S2": [], "S4": []}
    for w in wg:
        n = int(np.count_nonzero(w))
        ts = w[:n].astype(np.int64)
        per_it, head = 9, 5
        its = (n - head - 1) // per_it
        kernel_total.append(ts[n - 1] - ts[0])
        prologue.append(ts[1:head] - ts[0:head - 1])
        epilogue.append(ts[n - 1] - ts[n - 2])
        prev = ts[head - 1]
        for t in range(its):
            s = ts[head + t * per_it: head + (t + 1) * per_it]
            start = prev
            for k in range(4):
                if 0 < t < its - 1:
                    work[names[k]].append(s[2 * k] - prev)
                    wait[names[k]].append(s[2 * k + 1] - s[2 * k])
                    if names[k] in per_wave:
                        per_wave[names[k]].append((t, s[2 * k] - prev))
                prev = s[2 * k + 1]
            if 0 < t < its - 1:
                work["S5"].append(s[8] - prev)
            prev = s[8]
            (first_it if t == 0 else last_it if t == its - 1 else it_total).append(prev - start)
    for name, v in per_wave.items():
        by_it = {}
        for t, d in v:
            by_it.setdefault(t, []).append(d)
        for t, ds in by_it.items():
            duty[name] += sorted(ds)[-2:]
scale = (kernel_us / float(np.mean(kernel_total))) if kernel_us else 1.0
unit = "us" if kernel_us else "ticks"
us = lambda v: float(np.mean(v)) * scale
print(f"sampled waves: {len(kernel_total)}   scale {scale:.5f} {unit} per tick")
pro = np.mean(np.array(prologue), axis=0) * scale
print(f"kernel (first to last stamp): {us(kernel_total):8.2f} {unit}   iterations: first (input half only) {us(first_it):.2f}, middle {us(it_total):.2f}, last (late half only) {us(last_it):.2f}")
print(f"prologue: record / descriptors + tables {pro[0] + pro[1]:.2f}, first requests {pro[2]:.2f}, barrier {pro[3]:.2f}; epilogue {us(epilogue):.2f} {unit}")
tw = tb = 0.0
for n in names:
    w = us(work[n]); b = us(wait[n]) if n in wait else 0.0
    tw += w; tb += b
    print(f"  {n}: work {w:6.2f}   barrier wait {b:6.2f}   (max work {np.max(work[n]) * scale:6.2f})")
print(f"  middle iterations: work {tw:6.2f}, barrier wait {tb:6.2f} {unit}")
for name in ("S2", "S4"):
    if duty[name]:
        print(f"  {name}: the two wavefronts that ran the recurrences: {us(duty[name]):.2f} {unit} on average ({64 * 1000 * us(duty[name]) / 64 / 64:.1f} ns a step of 64)" if kernel_us else f"  {name}: duty wavefronts {us(duty[name]):.0f} ticks")
