"""Experiment (timing only, results racy on purpose): consecutive buffers launched on two alternating streams with nothing ordering them, to
see what the chip would gain if the tail of one steady-state launch could overlap with the head of the next (the ~6 us between
dependent launches of one stream)."""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
for k in range(6):
    b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
def run(tag, pick):
    torch.cuda.synchronize()
    for k in range(64): b.mix_device(frames, src.data_ptr(), dst.data_ptr(), stream=pick(k))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(400): b.mix_device(frames, src.data_ptr(), dst.data_ptr(), stream=pick(k))
    torch.cuda.synchronize()
    print(f"{tag}: step {(time.perf_counter() - t0) / 400 * 1e6:7.2f} us", flush=True)
run("one stream (ordered)", lambda k: None)
run("two streams alternating (unordered: racy, timing only)", lambda k: streams[k % 2].cuda_stream)
run("one stream (ordered)", lambda k: None)
