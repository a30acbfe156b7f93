#!/usr/bin/env python3
"""The host-pointer entry points on the headline workload: oalsfx_batch_mix (synchronous) and oalsfx_batch_mix_async (pipelined), the
latter with its copies through the runtime's copy engines (default) and as kernels (OALSFX_DEBUG_FLAGS 0x800000).  python scripts/host_io_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oalsfxpp_amd import desc, lib  # noqa: E402
from oalsfxpp_amd.api import Batch  # noqa: E402

n, F = 4096, 256
so = lib.load()
with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
    b.set_effect_type(0, desc.EAX_REVERB)
    b.apply_changes()
    xs = [b.pinned_array(F) for _ in range(3)]
    ys = [b.pinned_array(F) for _ in range(3)]
    for x in xs:
        x[:] = 0.1
    for k in range(6):
        b.mix(xs[0])
    t0 = time.perf_counter()
    for k in range(30):
        b._check(so.oalsfx_batch_mix(b._h, F, xs[k % 3].ctypes.data_as(lib._fp), ys[k % 3].ctypes.data_as(lib._fp)))
    print(f"oalsfx_batch_mix (synchronous, page-locked buffers):        {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step")
    for name, flags in (("copies through the runtime", 0), ("copies as kernels", 0x800000)):
        so.oalsfx_debug_set_flags(flags)
        for rep in range(2):
            for k in range(6):
                b.mix_async(xs[k % 3], ys[k % 3])
            b.wait()
            t0 = time.perf_counter()
            for k in range(60):
                b.mix_async(xs[k % 3], ys[k % 3])
            b.wait()
            dt = (time.perf_counter() - t0) / 60 * 1e3
        print(f"oalsfx_batch_mix_async, {name:28s}: {dt:.3f} ms per step  ({n * F / dt / 1e3:.0f} Msamples/s)")
    so.oalsfx_debug_set_flags(0)
