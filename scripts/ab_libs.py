#!/usr/bin/env python3
"""Same box, same process: two builds of liboalsfx_hip.so timed alternately on the headline workload (one batch per build, resident
together), many rounds; per round the average HIP-event duration of the steady-state reverb launch, then the median over rounds.

    python scripts/ab_libs.py <a.so> <b.so> [instances] [workload: eax | presets | type:<NAME of a ring-light type, e.g. type:CHORUS>] [frames per call] [--wall]

Raw ctypes on both libraries (two builds cannot share the Python mirror's single handle)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: one HIP runtime for everything)

from oalsfxpp_amd import desc  # noqa: E402

WALL = "--wall" in sys.argv   # time whole steps (host clock, k calls then one synchronisation) instead of the kernel's event pairs
sys.argv = [a for a in sys.argv if a != "--wall"]
paths = [os.path.abspath(p) for p in sys.argv[1:3]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
workload = sys.argv[4] if len(sys.argv) > 4 else "eax"
F = int(sys.argv[5]) if len(sys.argv) > 5 else 256   # frames per call
TYPE = getattr(desc, workload[5:]) if workload.startswith("type:") else desc.EAX_REVERB


class Lib:
    def __init__(self, path):
        so = self.so = C.CDLL(path)
        so.oalsfx_batch_create.restype = C.c_void_p
        so.oalsfx_batch_create.argtypes = [C.c_int] * 5
        for name, args in (("oalsfx_batch_destroy", [C.c_void_p]), ("oalsfx_batch_set_effect_type", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
                           ("oalsfx_batch_set_effect", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
                           ("oalsfx_batch_apply_changes", [C.c_void_p, C.c_int, C.c_int]), ("oalsfx_batch_synchronize", [C.c_void_p]),
                           ("oalsfx_batch_fill_synthetic", [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]),
                           ("oalsfx_batch_mix_device", [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
                           ("oalsfx_batch_kernel_timing", [C.c_void_p, C.c_int]),
                           ("oalsfx_batch_kernel_timing_read", [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
                           ("oalsfx_batch_event_overhead", [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
                           ("oalsfx_host_effect_defaults", [C.c_int, C.c_void_p]), ("oalsfx_host_preset", [C.c_int, C.c_void_p])):
            getattr(so, name).argtypes = args
        self.h = C.c_void_p(so.oalsfx_batch_create(n, desc.FMT_STEREO, 48000, 1, 0))
        assert self.h
        if workload == "presets":
            arr = (desc.Effect * n)()
            for i in range(n):
                so.oalsfx_host_effect_defaults(desc.EAX_REVERB, C.byref(arr[i]))
                so.oalsfx_host_preset(i % 113, C.byref(arr[i].props.reverb))
            assert so.oalsfx_batch_set_effect(self.h, 0, n, 0, arr, C.sizeof(desc.Effect))
        else:
            assert so.oalsfx_batch_set_effect_type(self.h, 0, n, 0, TYPE)
        assert so.oalsfx_batch_apply_changes(self.h, 0, n)
        self.src = [torch.empty(n * F * 2, dtype=torch.float32, device="cuda") for _ in range(4)]
        self.dst = torch.empty(n * F * 2, dtype=torch.float32, device="cuda")
        for k, s in enumerate(self.src):
            so.oalsfx_batch_fill_synthetic(self.h, F, k, C.c_void_p(s.data_ptr()), None)
        self.sync()
        self.run(8)
        self.sync()
        self.run(8)
        self.sync()

    def run(self, k):
        for i in range(k):
            assert self.so.oalsfx_batch_mix_device(self.h, F, C.c_void_p(self.src[i % 4].data_ptr()), C.c_void_p(self.dst.data_ptr()), None)

    def sync(self):
        assert self.so.oalsfx_batch_synchronize(self.h)

    def timed(self, k):
        if WALL:
            # whole steps on the host's clock (what bench.py's `value` is made of): k calls, one synchronisation
            import time
            self.sync()
            t0 = time.perf_counter()
            self.run(k)
            self.sync()
            return (time.perf_counter() - t0) / k * 1e6
        self.so.oalsfx_batch_kernel_timing(self.h, 1)
        self.run(k)
        self.sync()
        cnt, ms = C.c_int(0), C.c_double(0.0)
        self.so.oalsfx_batch_kernel_timing_read(self.h, TYPE, C.byref(cnt), C.byref(ms))
        g, gms = C.c_int(0), C.c_double(0.0)
        self.so.oalsfx_batch_kernel_timing_read(self.h, desc.REVERB + 16, C.byref(g), C.byref(gms))
        self.so.oalsfx_batch_kernel_timing(self.h, 0)
        return (ms.value + gms.value) / max(cnt.value, 1) * 1e3


# Where a batch's delay lines land in memory moves its launch time by a few per cent (the batch created second has been 3 % slower
# than the first, whichever build it belonged to): two batches per build, created in the order a b b a, and both counted.
order = (0, 1, 1, 0)
libs = [Lib(paths[w]) for w in order]
rounds = {0: [], 1: []}
per_batch = {k: [] for k in range(4)}
for rnd in range(12):
    for k in (0, 1, 2, 3) if rnd % 2 == 0 else (3, 2, 1, 0):
        libs[k].run(16 if F <= 512 else 4)
        libs[k].sync()
        t = libs[k].timed(64 if F <= 512 else 16)
        rounds[order[k]].append(t)
        per_batch[k].append(t)
pair = C.c_double(0.0)
if not WALL:
    libs[1].so.oalsfx_batch_event_overhead(libs[1].h, 200, C.byref(pair))
for which in (0, 1):
    v = sorted(rounds[which])
    print(f"{os.path.basename(paths[which]):36s} median of {len(v)} rounds x 64 launches (two batches): {v[len(v) // 2] - pair.value:6.2f} us   "
          f"(min {v[0] - pair.value:.2f}, max {v[-1] - pair.value:.2f}; empty event pair {pair.value:.2f} us taken off)")
print("per batch, in creation order: " + ", ".join(f"{os.path.basename(paths[order[k]])[12:-3] or 'product'} {sorted(per_batch[k])[len(per_batch[k]) // 2] - pair.value:.2f}" for k in range(4)))
for k in range(4):
    so = libs[k].so
    if hasattr(so, "oalsfx_debug_ring_address"):
        so.oalsfx_debug_ring_address.restype = C.c_ulonglong
        so.oalsfx_debug_ring_address.argtypes = [C.c_void_p, C.c_int, C.c_int]
        print(f"  batch {k}: rings at {so.oalsfx_debug_ring_address(libs[k].h, 0, 0):#x}, src {libs[k].src[0].data_ptr():#x} {libs[k].src[1].data_ptr():#x}, dst {libs[k].dst.data_ptr():#x}")
a, b = sorted(rounds[0])[len(rounds[0]) // 2] - pair.value, sorted(rounds[1])[len(rounds[1]) // 2] - pair.value
print(f"b / a = {b / a:.4f}")
for l in libs:
    l.so.oalsfx_batch_destroy(l.h)
