#!/usr/bin/env python3
"""Why is one of several batches of the headline workload 12 % faster than its siblings?  Four batches as scripts/ab_libs.py creates
them; each timed alone, then with another batch's input / output buffers.  python scripts/placement_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oalsfxpp_amd import desc, lib  # noqa: E402
from oalsfxpp_amd.api import Batch  # noqa: E402

n, F = 4096, 256
so = lib.load()
pre = os.environ.get("PROBE_PRE", "")
if pre:
    # allocator-state experiments: a large allocation before the first batch, kept ("keep") or freed again ("free")
    dummy = torch.empty(int(os.environ.get("PROBE_GIB", "4")) << 28, dtype=torch.float32, device="cuda")
    dummy.zero_()
    torch.cuda.synchronize()
    if pre == "free":
        del dummy
        torch.cuda.empty_cache()
B = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB)
    b.apply_changes()
    src = [torch.empty(n * F * 2, dtype=torch.float32, device="cuda") for _ in range(4)]
    dst = torch.empty(n * F * 2, dtype=torch.float32, device="cuda")
    for i, s in enumerate(src):
        b.fill_synthetic(F, i, s.data_ptr())
    b.synchronize()
    for _ in range(2):
        for i in range(8):
            b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
    B.append((b, src, dst))
pair = B[0][0].event_overhead(200)


def timed(b, src, dst, reps=3):
    out = []
    for _ in range(reps):
        for i in range(16):
            b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        b.kernel_timing(1)
        for i in range(64):
            b.mix_device(F, src[i % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        v = sorted(b.kernel_timing_samples(desc.EAX_REVERB))
        b.kernel_timing(0)
        out.append(v[len(v) // 2] - pair)
    return " ".join(f"{x:6.2f}" for x in out)


print("each batch with its own buffers (three medians of 64 launches):")
for k, (b, src, dst) in enumerate(B):
    print(f"  batch {k}: {timed(b, src, dst)}   rings {so.oalsfx_debug_ring_address(b._h, 0, 0):#x} src {src[0].data_ptr():#x} dst {dst.data_ptr():#x}")
print("each batch with the buffers of the batch after it:")
for k, (b, _, _) in enumerate(B):
    _, src, dst = B[(k + 1) % len(B)]
    print(f"  batch {k} + buffers of {(k + 1) % len(B)}: {timed(b, src, dst)}")
print("again with their own:")
for k, (b, src, dst) in enumerate(B):
    print(f"  batch {k}: {timed(b, src, dst)}")
for b, *_ in B:
    b.close()
