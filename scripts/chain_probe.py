"""Chained launches: step time of the headline loop with consecutive calls overlapping (default) and in plain stream order
(OALSFX_DEBUG=0x400), instances / frames from the command line.  python3 scripts/chain_probe.py [instances] [frames] [calls] [config3 | config4]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 400
mixed = len(sys.argv) > 4 and sys.argv[4] == "config4"   # BASELINE configs[3]: type 1 + i % 11, random properties: one mixed grid per step
chain4 = len(sys.argv) > 4 and sys.argv[4] == "config3"  # BASELINE configs[2]: chorus -> flanger -> echo -> EAX reverb, a step of two launches
b = Batch(n, desc.FMT_STEREO, 48000, 4 if chain4 else 1)
if chain4:
    for s, t in enumerate((desc.CHORUS, desc.FLANGER, desc.ECHO, desc.EAX_REVERB)): b.set_effect_type(s, t)
elif mixed:
    from oalsfxpp_amd.workloads import setup
    setup(b, "config4")
else:
    b.set_effect_type(0, desc.EAX_REVERB)
b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
for k in range(6):
    b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
for rep in range(3):
    b.synchronize(); torch.cuda.synchronize()
    before = b.chained_calls
    t0 = time.perf_counter()
    for k in range(calls): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = time.perf_counter() - t0
    same = b._lib.oalsfx_debug_chain_same_cu(b._h)
    print(f"{n} x {frames}: step {dt / calls * 1e6:7.2f} us, {b.chained_calls - before} of {calls} calls chained; hand-overs on one CU so far: {same} of {b.chained_calls * n}", flush=True)
