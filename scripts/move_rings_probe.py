#!/usr/bin/env python3
"""Is it the place of the delay lines that makes a batch fast or slow?  One batch of the headline workload, timed, its delay lines moved to a
fresh allocation (oalsfx_debug_move_rings), timed again, ... python scripts/move_rings_probe.py [moves] [keep_old]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oalsfxpp_amd import desc, lib  # noqa: E402
from oalsfxpp_amd.api import Batch  # noqa: E402

moves = int(sys.argv[1]) if len(sys.argv) > 1 else 8
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, F = 4096, 256
so = lib.load()
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
pair = b.event_overhead(200)


def timed():
    for i in range(16):
        b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
    b.synchronize()
    b.kernel_timing(1)
    for i in range(64):
        b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
    b.synchronize()
    v = sorted(b.kernel_timing_samples(desc.EAX_REVERB))
    b.kernel_timing(0)
    return v[len(v) // 2] - pair


for m in range(moves + 1):
    import ctypes as C
    t1, t2 = timed(), timed()
    us = C.c_double(0.0)
    so.oalsfx_debug_probe_rings(b._h, 32, C.byref(us))
    p1 = us.value
    so.oalsfx_debug_probe_rings(b._h, 32, C.byref(us))
    print(f"placement {m}: rings at {so.oalsfx_debug_ring_address(b._h, 0, 0):#x}: reverb kernel {t1:6.2f} {t2:6.2f} us   traffic-only probe {p1:6.2f} {us.value:6.2f} us", flush=True)
    if m < moves:
        assert so.oalsfx_debug_move_rings(b._h, keep)
b.close()
