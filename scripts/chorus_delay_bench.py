import os, sys, random, time
sys.path.insert(0, os.getcwd())
import torch
from oalsfxpp_amd import desc, workloads
from oalsfxpp_amd.api import Batch
sys.path.insert(0, "scripts")
F=256
def run(name, effects):
    n=len(effects)
    with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
        b.set_effect(0, effects); b.apply_changes()
        src=[torch.empty(n*F*2, dtype=torch.float32, device="cuda") for _ in range(4)]
        dst=torch.empty(n*F*2, dtype=torch.float32, device="cuda")
        for k,s in enumerate(src): b.fill_synthetic(F,k,s.data_ptr())
        for r in range(2):
            for k in range(16): b.mix_device(F, src[k%4].data_ptr(), dst.data_ptr())
            b.synchronize()
        t0=time.perf_counter()
        for k in range(200): b.mix_device(F, src[k%4].data_ptr(), dst.data_ptr())
        b.synchronize()
        print(f"{name:60s} {(time.perf_counter()-t0)/200*1e6:7.1f} us", flush=True)
E=workloads.make_effect
run("default chorus (triangle)", [E(desc.CHORUS)]*4096)
run("default chorus, sinusoid", [E(desc.CHORUS, waveform=0)]*4096)
rnd=[workloads.random_effect(random.Random(i), desc.CHORUS) for i in range(4096)]
run("random chorus", rnd)
def force(e, **kw):
    for k,v in kw.items(): setattr(e.props.chorus, k, v)
    return e
run("random chorus, all triangle", [force(workloads.random_effect(random.Random(i), desc.CHORUS), waveform=1) for i in range(4096)])
run("random chorus, delay >= 0.004", [force(e, delay=max(e.props.chorus.delay, 0.004)) for e in [workloads.random_effect(random.Random(i), desc.CHORUS) for i in range(4096)]])
run("random chorus, delay >= 0.004, triangle", [force(e, delay=max(e.props.chorus.delay, 0.004), waveform=1) for e in [workloads.random_effect(random.Random(i), desc.CHORUS) for i in range(4096)]])
for dl in (1,2,4,8,16,32):
    run(f"chorus delay {dl} samples depth 0 (triangle)", [E(desc.CHORUS, delay=dl/48000.0+1e-7, depth=0.0)]*4096)
