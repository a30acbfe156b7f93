#!/usr/bin/env python3
"""Where config 4's step time goes: its reverb instances alone (randomised properties), its ring-light instances alone, everything.
python scripts/config4_parts.py"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oalsfxpp_amd import desc, workloads  # noqa: E402
from oalsfxpp_amd.api import Batch  # noqa: E402

F = 256


def run(name, effects):
    n = len(effects)
    with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
        b.set_effect(0, effects)
        b.apply_changes()
        src = [torch.empty(n * F * 2, dtype=torch.float32, device="cuda") for _ in range(4)]
        dst = torch.empty(n * F * 2, dtype=torch.float32, device="cuda")
        for k, s in enumerate(src):
            b.fill_synthetic(F, k, s.data_ptr())
        for r in range(3):
            for k in range(16):
                b.mix_device(F, src[k % 4].data_ptr(), dst.data_ptr())
            b.synchronize()
        t0 = time.perf_counter()
        steps = 200
        for k in range(steps):
            b.mix_device(F, src[k % 4].data_ptr(), dst.data_ptr())
        b.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e6
        print(f"{name:58s} {n:5d} instances  step {dt:7.1f} us   plan {b.plan(0)}  {b.last_reverb_kernel}", flush=True)


all8192 = [workloads.random_effect(random.Random(i), workloads.config4_type(i)) for i in range(8192)]
reverbs = [e for e in all8192 if e.type in (desc.REVERB, desc.EAX_REVERB)]
lights = [e for e in all8192 if e.type not in (desc.REVERB, desc.EAX_REVERB)]
run("config 4 (8192 instances, 11 types, random properties)", all8192)
run("its reverbs alone", reverbs)
run("its ring-light instances alone", lights)
run("4096 random EAX reverbs", [workloads.random_effect(random.Random(i), desc.EAX_REVERB) for i in range(4096)])
e = workloads.make_effect(desc.EAX_REVERB, modulation_depth=0.5)
run("4096 default EAX reverbs with modulation depth 0.5", [e] * 4096)
e = workloads.make_effect(desc.EAX_REVERB, density=0.0)
run("4096 default EAX reverbs with density 0", [e] * 4096)
for t in range(1, 10):
    run(f"4096 random instances of type {desc.EFFECT_NAMES[t]}", [workloads.random_effect(random.Random(i), t) for i in range(4096)])
