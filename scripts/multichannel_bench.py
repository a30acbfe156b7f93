import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
for fmt, ch in ((desc.FMT_STEREO, 2), (desc.FMT_QUAD, 4), (desc.FMT_5POINT1, 6), (desc.FMT_7POINT1, 8)):
    n, frames = 4096, 256
    b = Batch(n, fmt, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
    src = torch.empty(n * frames * ch, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
    torch.cuda.synchronize()
    for _ in range(32): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    b.kernel_timing(4)
    t0 = time.perf_counter()
    for _ in range(200): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 200
    s = b.kernel_timing_read(desc.EAX_REVERB); g = b.kernel_timing_read(desc.REVERB + 16)
    print(f"channels {ch}: step {dt*1e6:7.1f} us  steady launches {s[0]} avg {s[1]/max(s[0],1)*1e3:6.1f} us  general launches {g[0]} avg {g[1]/max(g[0],1)*1e3:6.1f} us", flush=True)
    b.close()
