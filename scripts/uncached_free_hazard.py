"""The experiment behind the pool of uncached blocks (batch.cpp, UncachedPool): batches created and destroyed one after the other, mono
reverbs in front of stereo batches -- whose output then read back with 512 bytes of zeros at the start of a page, or whose reverbs went
wrong for a whole instance -- while uncached blocks went back to the runtime with hipFree as the runtime had handed them out (4 KiB
pieces among them).  Round 4's findings are in profiles/r04b_uncached_free_hazard/ and in the comment at UncachedPool: the buffer that
reads wrong is the batch's staging buffer in ordinary memory (MODE=device, the caller's own device buffers: never), a
hipDeviceSynchronize in front of the free does not help, whole 2 MiB granules do.  The pool now allocates in such granules only, so
  OALSFX_UNCACHED_POOL_MAX_GIB=0 TYPES=3,1,10 python3 scripts/uncached_free_hazard.py 1000 1000
(nothing waits for reuse: every block is freed when its batch goes) is today's form of the experiment, and comes out clean."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, random
import torch
from harness import OracleShadow, same_bits
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
from oalsfxpp_amd.workloads import random_effect
from oracle import oracle as orc
frames_list = [int(a) for a in sys.argv[1:]] or [2065, 64, 2065]
import os
for fmt in (desc.FMT_MONO, desc.FMT_STEREO):
    for t in [int(v) for v in os.environ.get("TYPES", "3,1,10").split(",")]:
        bad = 0
        for rep in range(6):
            rng = random.Random(100 * t + rep)
            n = 6
            with Batch(n, fmt, 48000, 1) as b:
                for i in range(n):
                    b.set_effect(0, random_effect(rng, t), first=i, count=1)
                b.apply_changes()
                sh = {i: OracleShadow(b, i) for i in range(n)}
                for k, f in enumerate(frames_list):
                    x = np.stack([orc.synth(7 + i, k, f * b.channels).reshape(f, b.channels) for i in range(n)])
                    if os.environ.get("MODE") == "device":   # (buffers of the caller's in device memory instead of the batch's staging buffers)
                        dx = torch.from_numpy(x).cuda(); dy = torch.full_like(dx, 7.0)
                        b.mix_device(f, dx.data_ptr(), dy.data_ptr()); b.synchronize(); y = dy.cpu().numpy()
                    else:
                        y = b.mix(x)
                    for i in range(n):
                        sh[i].last = sh[i].mix(x[i]); ok, nbad = same_bits(y[i], sh[i].last)
                        if not ok:
                            bad += 1
                            if bad <= 4:
                                ref = sh[i].last
                                d = np.nonzero((y[i].view(np.uint32) != ref.view(np.uint32)).any(axis=1))[0]
                                print("   rep", rep, "buffer", k, "instance", i, nbad, "samples; frames", d[0], "..", d[-1], len(d), "got", y[i][d[0]], "want", ref[d[0]], "in", x[i][d[0]])
        print(f"fmt {fmt} type {t} {desc.EFFECT_NAMES[t]}: {bad} bad buffers", flush=True)
