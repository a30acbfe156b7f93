"""Step time per effect type: 4096 instances of one type (default properties, stereo, 48 kHz, 256-frame buffers)."""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
names = {v: k for k, v in vars(desc).items() if k.isupper() and isinstance(v, int) and k in (
    "NULL", "CHORUS", "COMPRESSOR", "DEDICATED_DIALOG", "DEDICATED_LFE", "DISTORTION", "ECHO", "EQUALIZER", "FLANGER", "RING_MODULATOR", "REVERB", "EAX_REVERB")}
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
for t in sorted(names):
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect_type(0, t); b.apply_changes()
    for _ in range(16): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{names[t]:18s} step {dt*1e6:7.1f} us  {n*frames/dt/1e9:6.2f} Gsamples/s", flush=True)
    b.close()
