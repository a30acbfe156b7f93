"""Step time when a fraction of the instances gets a new preset before every buffer (4096 EAX reverbs, stereo, 256 frames).
  python3 scripts/update_storm_bench.py [k ...] [--per-call]
The changed instances are set with one `set_effect_at` call per buffer; --per-call: one `set_effect` call per instance, as the script
did until late in round 3 (forty foreign-function calls from Python cost 50 us)."""
import random, sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
presets = []
for i in range(113):
    e = lib.effect_defaults(desc.EAX_REVERB); e.props.reverb = lib.preset(i)[1]; presets.append(e)
rng = random.Random(1)
PER_CALL = "--per-call" in sys.argv
levels = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [0, 4, 40, 204, 1024]
def change(k):
    picks = rng.sample(range(n), k)
    if PER_CALL:
        for i in picks:
            b.set_effect(0, presets[rng.randrange(113)], first=i, count=1)
    elif k:
        b.set_effect_at(0, picks, [presets[rng.randrange(113)] for _ in picks])
for k in levels:
    for _ in range(8):   # untimed: the same update rate, so that staging buffers have their size
        change(k)
        if k: b.apply_changes()
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t_upd = 0.0
    t_mix = 0.0
    t0 = time.perf_counter()
    for step in range(50):
        u0 = time.perf_counter()
        change(k)
        if k: b.apply_changes()
        u1 = time.perf_counter()
        t_upd += u1 - u0
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
        t_mix += time.perf_counter() - u1
    b.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"{k:5d} updates per buffer: step {dt*1e6:8.1f} us (of which the setter calls from Python {t_upd/50*1e6:8.1f} us, the host inside mix_device {t_mix/50*1e6:8.1f} us)", flush=True)
