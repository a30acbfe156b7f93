import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
import numpy as np
for n in (4096,):
    t0=time.perf_counter()
    b=Batch(n, desc.FMT_STEREO, 48000, 1); b.set_effect_type(0, desc.EAX_REVERB); b.apply_changes()
    t1=time.perf_counter()
    src=torch.zeros(n*512, dtype=torch.float32, device="cuda"); dst=torch.zeros_like(src)
    b.mix_device(256, src.data_ptr(), dst.data_ptr()); b.synchronize()
    t2=time.perf_counter()
    print(f"n={n}: create+apply {t1-t0:.3f} s, first mix (allocation, placement search) {t2-t1:.3f} s, placement {b.placement()}")
    b.close()
