"""Tile-body time of a ring-light effect from the OALSFX_DEBUG_TIMELINE stamps (wave_instance: start, after init, then before and
after each tile's body).  usage: python scripts/timeline_wave.py <effect type number> [clock MHz]"""
import os, sys
sys.path.insert(0, ".")
os.environ["OALSFX_DEBUG_TIMELINE"] = "gpurun_out/timeline_wave.bin"
import numpy as np, torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
t = int(sys.argv[1]); mhz = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
n, frames = 4096, 256
b = Batch(n, desc.FMT_STEREO, 48000, 1)
b.set_effect_type(0, t); b.apply_changes()
src = torch.empty(n * frames * 2, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
for _ in range(8): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
b.synchronize(); b.close()
raw = np.fromfile("gpurun_out/timeline_wave.bin", dtype=np.uint64)[64 * 4 * 96:].reshape(64, 96)[:, :24]  # slot 0's stamps
rows = []
for w in raw:
    k = int(np.count_nonzero(w))
    if k < 4: continue
    ts = w[:k].astype(np.int64)
    body = [(ts[3 + 2 * i] - ts[2 + 2 * i]) for i in range((k - 2) // 2)]
    gaps = [(ts[4 + 2 * i] - ts[3 + 2 * i]) for i in range((k - 3) // 2)]
    rows.append((ts[1] - ts[0], np.mean(body), np.mean(gaps) if gaps else 0, ts[k - 1] - ts[0]))
r = np.array(rows) / mhz
print(f"type {t}: sampled {len(rows)} waves: init {r[:,0].mean():.2f} us, tile body {r[:,1].mean():.2f} us, between bodies {r[:,2].mean():.2f} us, first to last stamp {r[:,3].mean():.2f} us")
