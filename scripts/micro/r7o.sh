# 6.1 (seven channels a frame: one write-through store apiece) chained against stream order
mkdir -p gpurun_out/r7o
cat > /tmp/mc61_probe.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
for fmt, ch in ((desc.FMT_6POINT1, 7), (desc.FMT_5POINT1_REAR, 6)):
    n, frames = 4096, 256
    b = Batch(n, fmt, 48000, 1)
    b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
    src = torch.empty(n * frames * ch, device="cuda").uniform_(-1, 1); dst = torch.empty_like(src)
    for _ in range(8): b.mix_device(frames, src.data_ptr(), dst.data_ptr()); b.synchronize()
    for _ in range(32): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    before = b.chained_calls
    t0 = time.perf_counter()
    for _ in range(300): b.mix_device(frames, src.data_ptr(), dst.data_ptr())
    b.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(f"channels {ch}: step {dt*1e6:7.1f} us, {b.chained_calls - before} of 300 calls chained", flush=True)
    b.close()
PY
for rep in 1 2; do for flags in 0 0x40000; do
echo "== OALSFX_DEBUG_FLAGS=$flags"; OALSFX_DEBUG_FLAGS=$flags timeout -k 10 300 python /tmp/mc61_probe.py 2>/dev/null
done; done | tee gpurun_out/r7o/six_point_one.txt
