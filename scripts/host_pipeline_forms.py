#!/usr/bin/env python3
"""The pipelined host-pointer path (oalsfx_batch_mix_async) in its two forms on the headline workload, each fixed through
OALSFX_HOST_PIPELINE in a process of its own, beside the synchronous call and what the probe chooses by itself.
    python3 scripts/host_pipeline_forms.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time
sys.path.insert(0, %r)
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
n, F = 4096, 256
so = lib.load()
with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
    b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
    xs = [b.pinned_array(F) for _ in range(3)]; ys = [b.pinned_array(F) for _ in range(3)]
    for x in xs: x[:] = 0.1
    for k in range(6): b.mix(xs[0])
    t0 = time.perf_counter()
    for k in range(30): b._check(so.oalsfx_batch_mix(b._h, F, xs[k %% 3].ctypes.data_as(lib._fp), ys[k %% 3].ctypes.data_as(lib._fp)))
    sync_ms = (time.perf_counter() - t0) / 30 * 1e3
    for k in range(52): b.mix_async(xs[k %% 3], ys[k %% 3])
    b.wait()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(60): b.mix_async(xs[k %% 3], ys[k %% 3])
        b.wait()
        best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
    print("synchronous %%.3f ms per step; pipelined %%.3f ms per step; form, probe us (three streams, one stream): %%s" %% (sync_ms, best, b.host_pipeline()))
''' % ROOT
for form in ("3", "1", ""):
    env = dict(os.environ)
    if form: env["OALSFX_HOST_PIPELINE"] = form
    else: env.pop("OALSFX_HOST_PIPELINE", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(("OALSFX_HOST_PIPELINE=" + form if form else "probe") + ":", (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1], flush=True)
