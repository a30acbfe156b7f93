#!/usr/bin/env python3
"""Where in the card's memory does the reverb's ring traffic run fast?  Allocates chunk after chunk (each the delay lines of `instances`
EAX reverbs, all kept) until `gib` GiB are taken, and times the traffic-only probe (k_stream_pattern) on every one.
python scripts/vram_map.py [gib] [instances]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oalsfxpp_amd import lib  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
slab = 235520
so = lib.load()
chunks, taken = [], 0.0
print(torch.cuda.get_device_name(0), f"free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB")
while taken + n * slab * 4 / 2**30 <= gib:
    t = torch.zeros(n * slab, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    us = C.c_double(0.0)
    assert so.oalsfx_debug_probe_pointer(C.c_void_p(t.data_ptr()), n, slab, 16, C.byref(us))
    chunks.append(t)
    taken += n * slab * 4 / 2**30
    print(f"chunk {len(chunks) - 1:3d}  at {t.data_ptr():#x}  {taken:7.1f} GiB taken so far: {us.value:6.2f} us per launch", flush=True)
