"""256-frame calls of the headline loop after one odd-sized call: the delay lines' write position is then no longer on a cache-line
boundary (a line holds 8 frames of a 4-line ring) and stays off it for good.  python3 scripts/misaligned_bench.py [first call sizes]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
firsts = [int(a) for a in sys.argv[1:]] or [0, 441, 100, 37]
for first in firsts:
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
    src = torch.empty(n * 512 * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
    if first: b.mix_device(first, src.data_ptr(), dst.data_ptr())
    for k in range(6):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
    for rep in range(3):
        b.synchronize(); t0 = time.perf_counter()
        for k in range(400): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
        b.synchronize(); dt = (time.perf_counter() - t0) / 400
    print(f"256-frame calls after a first call of {first} frames: step {dt*1e6:.2f} us ({b.chained_calls} chained)", flush=True)
    b.close()
