"""Step time for call sizes that are not whole 64-frame tiles (4096 EAX reverbs, stereo); OALSFX_DEBUG_FLAGS=8 forces the
general kernel alone for comparison."""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n = 4096
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
for frames in (256, 480, 441, 512, 1024, 960):
    src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
    torch.cuda.synchronize()
    for _ in range(16): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print(f"frames {frames:5d}: step {dt*1e6:7.1f} us  {n*frames/dt/1e9:6.2f} Gsamples/s", flush=True)
