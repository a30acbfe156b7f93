"""Where a fused run of ring-light slots spends its time (config 3: chorus, flanger, echo in one launch, one wavefront per instance
walking its slots): per slot the start-up (type read, state loads), the tile bodies, the parts between them, and the gap to the next
slot, from the OALSFX_DEBUG_TIMELINE stamps.  usage: python scripts/timeline_slots.py [clock MHz]"""
import os, sys
sys.path.insert(0, ".")
os.environ["OALSFX_DEBUG_TIMELINE"] = "gpurun_out/timeline_slots.bin"
import numpy as np, torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
mhz = float(sys.argv[1]) if len(sys.argv) > 1 else 2100.0
n, frames = 4096, 256
b = Batch(n, desc.FMT_STEREO, 48000, 4)
for slot, t in enumerate((desc.CHORUS, desc.FLANGER, desc.ECHO, desc.EAX_REVERB)):
    b.set_effect_type(slot, t)
b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
for _ in range(8): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
b.synchronize(); b.close()
raw = np.fromfile("gpurun_out/timeline_slots.bin", dtype=np.uint64)[64 * 4 * 96:].reshape(64, 4, 24).astype(np.int64)
rows = [w for w in raw if w[0, 0] != 0]
t0 = np.mean([w[0, 0] for w in rows])
print(f"sampled {len(rows)} wavefronts; microseconds at {mhz:.0f} MHz")
prev_end = None
for s in range(3):
    k = 2 + 2 * (frames // 64)   # start, after init, then before and after each tile's body
    st = np.array([w[s][:k] for w in rows])
    d = np.diff(st, axis=1).mean(axis=0) / mhz
    start = (st[:, 0].mean() - t0) / mhz
    gap = "" if prev_end is None else f", {start - prev_end:.2f} after the slot before ended"
    print(f"slot {s}: first stamp at {start:6.2f}{gap}; init {d[0]:.2f}; then (between, body) x tiles: " + " ".join(f"({d[1 + 2 * i]:.2f}, {d[2 + 2 * i]:.2f})" for i in range((k - 2) // 2)))
    prev_end = (st[:, k - 1].mean() - t0) / mhz
