"""Long runs of the chained shapes of round 4 at full size, every buffer of the followed instances against the oracle: 400 steps of two
launches (4096 instances, presets in the reverbs' slot) and 300 of the mixed grid (8192 instances, BASELINE configs[3]), no
synchronisation inside a run.  python3 scripts/long_chained_runs.py   (through gpurun; uses the helpers of tests/test_gpu_chained.py)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
torch.cuda.init()   # (before the library touches the device: the other order leaves torch without one on this stack)
import test_gpu_chained as t
from oalsfxpp_amd import desc
t0 = time.perf_counter()
t._a_run_of_steps_of_two_launches(4096, desc.FMT_STEREO, 91000, [256] * 400, 390)
print(f"400 steps of two launches, 4096 instances: every buffer of the followed instances, states and delay lines match ({time.perf_counter() - t0:.0f} s)", flush=True)
t0 = time.perf_counter()
t._a_run_of_one_mixed_grid(8192, desc.FMT_STEREO, 92000, [256] * 300, 290, workload="config4")
print(f"300 steps of the mixed grid, 8192 instances: every buffer of the followed instances, states and delay lines match ({time.perf_counter() - t0:.0f} s)", flush=True)
