"""Ring-light effect batches as chained launches against plain stream order (OALSFX_DEBUG_FLAGS 0x400), same process, same box:
4096 instances of one effect type (and of all ten in one slot, and three slots of them), 256-frame calls, microseconds per step.
python3 scripts/light_chain_bench.py [instances] [frames]"""
import sys, time
sys.path.insert(0, ".")
import random
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
from oalsfxpp_amd.workloads import random_effect

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
so = lib.load()
TYPES = [("chorus", desc.CHORUS), ("flanger", desc.FLANGER), ("echo", desc.ECHO), ("equalizer", desc.EQUALIZER), ("distortion", desc.DISTORTION),
         ("ring modulator", desc.RING_MODULATOR), ("compressor", desc.COMPRESSOR), ("dedicated", desc.DEDICATED_DIALOG)]


def step_us(b, calls=400):
    src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1)
    dst = torch.empty_like(src)
    for _ in range(8):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    out = []
    for flags in (0x80, 0x400, 0x80, 0x400):   # (0x80: batches without any reverb chain too -- the product leaves them in stream order)
        so.oalsfx_debug_set_flags(flags)
        for _ in range(20):
            b.mix_device(frames, src.data_ptr(), dst.data_ptr())
        b.synchronize()
        before = b.chained_calls
        t0 = time.perf_counter()
        for _ in range(calls):
            b.mix_device(frames, src.data_ptr(), dst.data_ptr())
        b.synchronize()
        out.append(((time.perf_counter() - t0) / calls * 1e6, b.chained_calls - before))
    so.oalsfx_debug_set_flags(0)
    return out


def report(name, b):
    r = step_us(b)
    print(f"{name:34s} chained {r[0][0]:6.2f} / {r[2][0]:6.2f} us ({r[0][1]} of 400 calls chained)   stream order {r[1][0]:6.2f} / {r[3][0]:6.2f} us", flush=True)


for name, t in TYPES:
    with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
        b.set_effect_type(0, t)
        b.apply_changes()
        report(f"{n} x {name}", b)
rng = random.Random(5)
with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
    b.set_effect(0, [random_effect(rng, TYPES[i % len(TYPES)][1]) for i in range(n)])
    b.apply_changes()
    report(f"{n} x eight types in one slot", b)
for share in (64, 16, 4):
    with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
        b.set_effect(0, [lib.effect_defaults(desc.EAX_REVERB) if i % share == 0 else random_effect(rng, TYPES[i % len(TYPES)][1]) for i in range(n)])
        b.apply_changes()
        src = torch.zeros(n * frames * 2, device="cuda"); dst = torch.empty_like(src)
        for _ in range(6):
            b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()   # through the reverbs' start-up; proven
        report(f"{n} x eight types, 1 in {share} an EAX reverb", b)
with Batch(n, desc.FMT_STEREO, 48000, 3) as b:
    for s, t in enumerate((desc.CHORUS, desc.FLANGER, desc.ECHO)):
        b.set_effect_type(s, t)
    b.apply_changes()
    report(f"{n} x chorus -> flanger -> echo", b)
