import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n = 4096
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
src = torch.empty(n * 2048 * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
for _ in range(8): b.mix_device(256, src.data_ptr(), dst.data_ptr())
for frames in (16, 32, 48, 64, 96):
    for _ in range(32): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"frames {frames:4d}: step {dt*1e6:7.1f} us  {n*frames/dt/1e9:6.2f} Gsamples/s", flush=True)
