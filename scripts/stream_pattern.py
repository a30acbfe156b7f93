"""What the memory system sustains for the reverb kernel's ring traffic alone (see k_stream_pattern), and plain
streaming ceilings of the box (torch copy / fill / sum of 1 GiB)."""
import ctypes as C
import sys
import time
sys.path.insert(0, ".")
import torch
from oalsfxpp_amd import lib
so = lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
gib = a.numel() * 4
print(f"copy 1 GiB: {2 * gib / timed(lambda: b.copy_(a)) / 1e12:.2f} TB/s (read + write)   fill: {gib / timed(lambda: b.fill_(1.0)) / 1e12:.2f} TB/s   "
      f"sum: {gib / timed(lambda: a.sum()) / 1e12:.2f} TB/s", flush=True)
del a, b
mb = n * 256 * 48 * 4 / 1e6
for slab, skew in ((235520, 0), (235520, 64), (235520, 97), (235520 + 1024, 0), (235520 + 64, 0), (235520 + 32, 0), (262144, 0), (235520 + 3072, 0)):
    for v in (1, 4):
        us = C.c_double()
        assert so.oalsfx_debug_stream_pattern(0, n, v, 200, slab, skew, C.byref(us))
        print(f"instances {n}  slab {slab} floats  skew {skew:3d}  burst {256 * v:5d} B  {us.value:7.2f} us/launch  {mb / us.value:6.3f} TB/s algorithmic", flush=True)
