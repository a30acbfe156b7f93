# 6.1 chained with its stores paired by the frame's parity (OALSFX_CHAIN_ODD=1: experiment) against stream order
mkdir -p gpurun_out/r7r
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
for rep in 1 2; do for odd in 0 1; do
if [ $odd = 1 ]; then export OALSFX_CHAIN_ODD=1; else unset OALSFX_CHAIN_ODD; fi
echo "== OALSFX_CHAIN_ODD=$odd"; timeout -k 10 300 python /tmp/mc61_probe.py 2>/dev/null | grep "channels 7"
done; done | tee gpurun_out/r7r/six_point_one_paired.txt
OALSFX_CHAIN_ODD=1 timeout -k 10 300 python -m pytest tests/test_gpu_chained.py -q -k "more_than_two_channels_match" 2>&1 | tail -5
