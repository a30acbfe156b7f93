#!/usr/bin/env python3
"""Same physical memory, different slab strides: does the layout decide how fast the ring traffic runs, or the place?
Allocates a few 6 GiB buffers and times the placement probe over 4096 slabs laid out with different strides in each.
python scripts/stride_probe.py [buffers]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oalsfxpp_amd import lib  # noqa: E402

so = lib.load()
n = 4096
base = 235520
strides = [base, base + 64, base + 1024, base + 1024 * 3, base + 1024 * 7, base + 4096 * 5, 262144, 262144 + 1024, 300000 // 64 * 64, 393216]
bufs = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    t = torch.zeros(6 << 28, dtype=torch.float32, device="cuda")  # 6 GiB
    torch.cuda.synchronize()
    bufs.append(t)
    out = []
    for st in strides:
        us = C.c_double(0.0)
        assert so.oalsfx_debug_probe_pointer(C.c_void_p(t.data_ptr()), n, st, 12, C.byref(us))
        out.append(f"{st}:{us.value:5.1f}")
    print(f"buffer {k} at {t.data_ptr():#x}: " + "  ".join(out), flush=True)
