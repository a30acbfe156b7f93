#!/usr/bin/env python3
"""What evaluating the two first-order T60 sections of the (EAX) reverb as a wavefront parallel prefix would do to the results
(BASELINE.json north_star suggests it): every EFX preset, stereo, 48 kHz, 64 buffers of 256 frames of the benchmark's
noise, the bit-exact oracle against the same source with those sections composed as affine maps in six doubling steps of fp32
arithmetic (oracle/liboracle_ps.so, `make -C oracle ps`).  CPU only.  Prints the parity metric of SURVEY 7 -- max |a - b| over
max(peak, 1) -- per preset class and overall."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import OracleApi, preset_effect  # noqa: E402
from oalsfxpp_amd import desc, lib  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ps = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_ps.so"))
base = orc.oracle_lib()
for name in ("oracle_create", "oracle_destroy", "oracle_set_source", "oracle_set_slot", "oracle_mix"):
    getattr(ps, name).restype = getattr(base, name).restype
    getattr(ps, name).argtypes = getattr(base, name).argtypes

worst, rows = 0.0, []
F, BUFFERS = 256, 64
for i in range(lib.preset_count()):
    api = OracleApi(desc.FMT_STEREO, 48000, 1)
    api.set_effect(0, preset_effect(i))
    api.apply_changes()
    api.refresh()
    h = C.c_void_p(ps.oracle_create(2, 1))
    ps.oracle_set_source(h, C.byref(api.source_params))
    ps.oracle_set_slot(h, 0, C.byref(api.params[0]), 1)
    err = peak = 0.0
    for k in range(BUFFERS):
        x = orc.synth(i, k, F * 2)
        a = api.oracle.mix(x.reshape(F, 2)).reshape(-1)
        b = np.empty_like(x)
        ps.oracle_mix(h, F, x.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)))
        err = max(err, float(np.max(np.abs(a - b))))
        peak = max(peak, float(np.max(np.abs(a))))
    ps.oracle_destroy(h)
    rel = err / max(peak, 1.0)
    rows.append((rel, err, peak, lib.preset(i)[0]))
    worst = max(worst, rel)
blocks, used = C.c_long(0), C.c_long(0)
ps.oracle_ps_counts(C.byref(blocks), C.byref(used))
rows.sort(reverse=True)
print(f"{len(rows)} presets x {BUFFERS} buffers of {F} frames; {used.value} of {blocks.value} reverb blocks took the parallel-prefix evaluation")
print(f"worst max|diff| / max(peak, 1): {worst:.3e}   (tolerance of the task: 1e-5)")
for rel, err, peak, name in rows[:8]:
    print(f"  {name:28s} max|diff| {err:.3e}  peak {peak:7.3f}  relative {rel:.3e}")
print(f"  median preset: {rows[len(rows) // 2][0]:.3e}")
