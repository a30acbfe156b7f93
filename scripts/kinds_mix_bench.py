"""Step time of 4096 EAX reverbs of which M use another preset than the rest (two kinds of steady instances in one grid):
    python3 scripts/kinds_mix_bench.py <other preset> M [M ...]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
other = int(sys.argv[1])
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
e = lib.effect_defaults(desc.EAX_REVERB); e.props.reverb = lib.preset(other)[1]
for m in [int(a) for a in sys.argv[2:] if not a.startswith("--")]:
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB)
    if m and "--interleave" in sys.argv:
        step = n // m
        for i in range(0, n, step): b.set_effect(0, e, first=i, count=1)
    elif m: b.set_effect(0, e, first=n - m if "--tail" in sys.argv else 0, count=m)
    b.apply_changes()
    for k in range(6):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
    for k in range(64):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for k in range(300):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(f"{m:5d} instances of preset {other} among {n}: step {dt*1e6:7.1f} us  plan {b.plan(0)}  {b.last_reverb_kernel}", flush=True)
    b.close()
