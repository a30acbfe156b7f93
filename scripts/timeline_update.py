"""Leaves the general reverb path's phase stamps (OALSFX_DEBUG_TIMELINE file) for a buffer right after a parameter update:
the sampled instances (every 64th) get a new preset before every buffer.  Read the file with scripts/timeline_general.py."""
import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("OALSFX_DEBUG_TIMELINE", "gpurun_out/timeline_update.bin")
import torch
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
n, frames = int(os.environ.get('N', '4096')), 256
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
presets = []
for i in range(113):
    e = lib.effect_defaults(desc.EAX_REVERB); e.props.reverb = lib.preset(i)[1]; presets.append(e)
for step in range(40):
    if step >= 32:
        for i in range(0, n, 64 if len(sys.argv) < 2 else int(sys.argv[1])):
            b.set_effect(0, presets[(step * 7 + i) % 113], first=i, count=1)
        b.apply_changes()
    b.mix_device(frames, src.data_ptr(), dst.data_ptr())
b.synchronize()
b.close() if hasattr(b, "close") else None
del b
