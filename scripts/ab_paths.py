#!/usr/bin/env python3
"""Same box, same process: two code paths of the library (OALSFX_DEBUG_FLAGS bits, oalsfx_debug_set_flags) timed alternately on two
batches of the headline workload, many rounds, medians.  python scripts/ab_paths.py [flagsA] [flagsB] [instances]
Default: A = 0x200000 (proven instances through the believing builds, i.e. round 1's kernel), B = 0 (proven-steady builds)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: one HIP runtime for both)

from oalsfxpp_amd import desc, lib  # noqa: E402
from oalsfxpp_amd.api import Batch  # noqa: E402

fa = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x200000
fb = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
F = 256
so = lib.load()


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


batches = []
for flags in (fa, fb):
    so.oalsfx_debug_set_flags(flags)   # in force while the batch settles: what ends up in the hot records (tap tables) is built under them
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB)
    b.apply_changes()
    src = [torch.empty(n * F * 2, dtype=torch.float32, device="cuda") for _ in range(4)]
    dst = torch.empty(n * F * 2, dtype=torch.float32, device="cuda")
    for k, s in enumerate(src):
        b.fill_synthetic(F, k, s.data_ptr())
    b.synchronize()
    for _ in range(3):
        for k in range(8):
            b.mix_device(F, src[k % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
    batches.append((b, src, dst))
res = {0: [], 1: []}
names = {}
for rnd in range(12):
    for which, flags in ((0, fa), (1, fb)):
        so.oalsfx_debug_set_flags(flags)
        b, src, dst = batches[which]
        for k in range(24):
            b.mix_device(F, src[k % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        b.kernel_timing(1)
        for k in range(64):
            b.mix_device(F, src[k % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        res[which] += b.kernel_timing_samples(desc.EAX_REVERB)
        names[which] = b.last_reverb_kernel
        b.kernel_timing(0)
pair = batches[1][0].event_overhead(200)
for which, flags in ((0, fa), (1, fb)):
    v = sorted(res[which])
    print(f"flags {flags:#x}: {names[which]}: median {med(v) - pair:.2f} us, p10 {v[len(v) // 10] - pair:.2f}, p90 {v[len(v) * 9 // 10] - pair:.2f}, "
          f"mean {sum(v) / len(v) - pair:.2f} ({len(v)} launches, empty event pair {pair:.2f} us taken off)")
for b, _, _ in batches:
    b.close()
