#!/usr/bin/env python3
"""Does it matter where a batch's delay lines land?  Several batches of the headline workload created one after the other in one
process, each timed (median HIP-event duration of the steady-state kernel over 3 x 64 launches, round robin), with the device
address of its first slab.  python scripts/placement_bench.py [batches] [instances]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oalsfxpp_amd import desc, lib  # noqa: E402
from oalsfxpp_amd.api import Batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
F = 256
so = lib.load()
items = []
for k in range(count):
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB)
    b.apply_changes()
    src = [torch.empty(n * F * 2, dtype=torch.float32, device="cuda") for _ in range(4)]
    dst = torch.empty(n * F * 2, dtype=torch.float32, device="cuda")
    for i, s in enumerate(src):
        b.fill_synthetic(F, i, s.data_ptr())
    for _ in range(3):
        for i in range(8):
            b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
    items.append((b, src, dst, []))
for rnd in range(3):
    for b, src, dst, samples in items:
        for i in range(16):
            b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        b.kernel_timing(1)
        for i in range(64):
            b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        samples += b.kernel_timing_samples(desc.EAX_REVERB)
        b.kernel_timing(0)
pair = items[0][0].event_overhead(200)
for k, (b, src, dst, samples) in enumerate(items):
    v = sorted(samples)
    a0 = so.oalsfx_debug_ring_address(b._h, 0, 0)
    a1 = so.oalsfx_debug_ring_address(b._h, 1, 0)
    print(f"batch {k}: slab 0 at {a0:#014x} (mod 2 MiB {a0 % (2 << 20):#09x}, mod 1 GiB {a0 % (1 << 30):#011x}), stride {a1 - a0}, "
          f"src {src[0].data_ptr():#014x} dst {dst.data_ptr():#014x}: median {v[len(v) // 2] - pair:6.2f} us  (p10 {v[len(v) // 10] - pair:.2f}, p90 {v[len(v) * 9 // 10] - pair:.2f})")
for b, *_ in items:
    b.close()
