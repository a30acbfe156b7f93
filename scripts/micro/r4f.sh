mkdir -p gpurun_out/r4f
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py tests/test_gpu_parity.py tests/test_gpu_crossfade.py -q -m gpu > gpurun_out/r4f/tests.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/r4f/tests.log
echo "== ragged sizes, product"; timeout -k 10 300 python3 scripts/ragged_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/ragged.txt
echo "== ragged sizes, 0x2000 (as round 3)"; OALSFX_DEBUG_FLAGS=0x2000 timeout -k 10 300 python3 scripts/ragged_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4f/ragged.txt
