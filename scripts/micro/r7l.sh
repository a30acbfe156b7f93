# more than two channels chained against stream order as first measured (OALSFX_CHAIN_MC was the experiment's switch; chaining is the default since, OALSFX_DEBUG_FLAGS=0x40000 the way back): quad / 5.1 / 7.1, 4096 EAX reverbs, 256-frame calls
mkdir -p gpurun_out/r7l
cat > /tmp/mc_probe.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
for fmt, ch in ((desc.FMT_QUAD, 4), (desc.FMT_5POINT1, 6), (desc.FMT_7POINT1, 8)):
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
for rep in 1 2; do for mc in 0 1; do
if [ $mc = 1 ]; then export OALSFX_CHAIN_MC=1; else unset OALSFX_CHAIN_MC; fi
echo "== OALSFX_CHAIN_MC=$mc"; timeout -k 10 300 python /tmp/mc_probe.py 2>/dev/null
done; done | tee gpurun_out/r7l/multichannel_chained.txt
OALSFX_CHAIN_MC=1 timeout -k 10 600 python -m pytest tests/ -q -m gpu -k "channel or format or quad or surround" 2>&1 | tail -3
