"""Step time of 4096 EAX reverbs cycling through the presets of one kind (or of several kinds), to see what each build of the steady-state
kernel makes of its own kind of presets:  python3 scripts/kinds_presets_bench.py"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
n, frames = 4096, 256
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
def effect(i):
    e = lib.effect_defaults(desc.EAX_REVERB); e.props.reverb = lib.preset(i)[1]; return e
def kind(i):
    p = lib.derive_slot(desc.FMT_STEREO, 48000, lib.effect_normalized(effect(i))).u.reverb
    lo = min(min(p.early_tap), min(p.early_ap_off), min(p.early_line_off), min(x - p.late_feed_tap for x in p.late_tap), min(p.late_ap_off), min(p.late_line_off))
    return 2 if (lo < 64 or p.mod_depth != 0.0) else 1 if lo < 128 else 0
kinds = {k: [i for i in range(113) if kind(i) == k] for k in range(3)}
print({k: len(v) for k, v in kinds.items()}, flush=True)
def run(name, presets):
    b = Batch(n, desc.FMT_STEREO, 48000, 1)
    b.set_effect(0, [effect(presets[i % len(presets)]) for i in range(n)])
    b.apply_changes()
    for k in range(6):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
    for k in range(64):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    t0 = time.perf_counter()
    for k in range(300):
        b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(f"{name:28s}: step {dt*1e6:7.1f} us  plan {b.plan(0)}  {b.last_reverb_kernel}", flush=True)
    b.close()
run("kind 0 presets only", kinds[0])
run("kind 1 presets only", kinds[1])
run("kind 2 presets only", kinds[2])
run("kinds 0 + 1", kinds[0] + kinds[1])
run("kinds 0 + 2", kinds[0] + kinds[2])
run("all (i % 113)", list(range(113)))
for i in kinds[0][:40]:
    if "--each" in sys.argv: run(f"preset {i} ({lib.preset(i)[0]})", [i])
