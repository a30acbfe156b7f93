"""Step time per 64-frame tile for whole-tile call sizes (4096 EAX reverbs, stereo): what a launch costs beyond its tiles."""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n = 4096
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
for frames in (256, 64, 128, 256, 512, 1024, 2048, 256):
    src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
    torch.cuda.synchronize()
    reps = max(20, 25600 // frames)
    for _ in range(max(4, 2048 // frames)): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"frames {frames:5d}: step {dt*1e6:7.1f} us = {dt*1e6/(frames/64):6.2f} us per tile, {n*frames/dt/1e9:6.2f} Gsamples/s", flush=True)
