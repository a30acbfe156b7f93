"""Does the chip overlap the prologue of one launch with the tile loop of another?  The default workload (4096 EAX reverbs,
stereo, 256 frames) as one batch, and as 2 / 4 independent sub-batches on their own streams, enqueued round-robin."""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n, frames, steps = 4096, 256, 400
for parts in (1, 2, 1, 2, 1, 2, 3, 1, 2):
    m = (n // parts + 3) // 4 * 4
    bs = [Batch(m, desc.FMT_STEREO, 48000, 1) for _ in range(parts)]
    src = [torch.empty(m * frames * 2, device="cuda").uniform_(-1, 1) for _ in range(parts)]
    dst = [torch.empty_like(s) for s in src]
    for b in bs:
        b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
    for _ in range(64):
        for k, b in enumerate(bs): b.mix_device(frames, src[k].data_ptr(), dst[k].data_ptr())
    for b in bs: b.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for k, b in enumerate(bs): b.mix_device(frames, src[k].data_ptr(), dst[k].data_ptr())
    t_host = (time.perf_counter() - t0) / steps
    for b in bs: b.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{parts} sub-batches of {m}: {dt*1e6:7.1f} us per buffer of all {n} instances = {n*frames/dt/1e9:6.2f} Gsamples/s (host enqueue {t_host*1e6:5.1f} us)", flush=True)
    del bs
