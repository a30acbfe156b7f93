"""Step time when a few of 4096 EAX reverbs get a send change before every buffer: they stay steady, but the host has to wait for the
device to confirm them again, and meanwhile the slot is not all proven."""
import sys, time, random
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
rng = random.Random(3)
for k in (0, 4):
    for _ in range(16):
        for i in rng.sample(range(n), k): b.set_send_props(-1, rng.uniform(0.5, 1.0), 1.0, 1.0, first=i, count=1)
        if k: b.apply_changes()
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        for i in rng.sample(range(n), k): b.set_send_props(-1, rng.uniform(0.5, 1.0), 1.0, 1.0, first=i, count=1)
        if k: b.apply_changes()
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    print(f"{k} send changes per buffer: step {(time.perf_counter()-t0)/100*1e6:.1f} us, plan {b.plan(0)}", flush=True)
