"""Steps of two launches with the reverbs' slot holding EFX preset i % 113 (a grid of several kinds) instead of defaults: chained against
stream order (OALSFX_DEBUG_FLAGS 0x400), same process.  python3 scripts/two_launch_presets_bench.py [instances]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
sys.path.insert(0, "tests")
from harness import preset_effect

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = 256
so = lib.load()
for label, effects in (("defaults", None), ("presets", [preset_effect(i % 113) for i in range(n)])):
    with Batch(n, desc.FMT_STEREO, 48000, 4) as b:
        for s, t in enumerate((desc.CHORUS, desc.FLANGER, desc.ECHO)):
            b.set_effect_type(s, t)
        if effects:
            b.set_effect(3, effects)
        else:
            b.set_effect_type(3, desc.EAX_REVERB)
        b.apply_changes()
        src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1)
        dst = torch.empty_like(src)
        for _ in range(8):
            b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
        out = []
        for flags in (0, 0x400, 0, 0x400):
            so.oalsfx_debug_set_flags(flags)
            for _ in range(30):
                b.mix_device(frames, src.data_ptr(), dst.data_ptr())
            b.synchronize()
            before = b.chained_calls
            t0 = time.perf_counter()
            for _ in range(300):
                b.mix_device(frames, src.data_ptr(), dst.data_ptr())
            b.synchronize()
            out.append(((time.perf_counter() - t0) / 300 * 1e6, b.chained_calls - before))
        so.oalsfx_debug_set_flags(0)
        print(f"{n} x (chorus, flanger, echo, EAX reverb {label}): chained {out[0][0]:6.1f} / {out[2][0]:6.1f} us ({out[0][1]} of 300 chained, {b.last_reverb_kernel[:28]})   stream order {out[1][0]:6.1f} / {out[3][0]:6.1f} us", flush=True)
