import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
def run(tag):
    torch.cuda.synchronize()
    for _ in range(16): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    print(f"{tag}: step {(time.perf_counter() - t0) / 100 * 1e6:7.1f} us", flush=True)
run("no filters")
b.set_send_props(-1, 1.0, 0.5, 1.0); b.apply_changes()
run("direct high-shelf on every instance")
b.set_send_props(0, 1.0, 0.5, 0.5); b.apply_changes()
run("direct + aux (both shelves)")
b.set_send_props(-1, 1.0, 1.0, 1.0); b.set_send_props(0, 1.0, 1.0, 1.0); b.set_send_props(0, 1.0, 0.5, 1.0, first=0, count=1); b.apply_changes()
run("one instance filtered")
