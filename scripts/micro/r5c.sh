mkdir -p gpurun_out/r5c
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5c/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5c/tests.log; grep -E "^E  " gpurun_out/r5c/tests.log | head
for rep in 1 2; do
echo "== ragged sizes: line-aligned stores off (0x4000)"; OALSFX_DEBUG_FLAGS=0x4000 timeout -k 10 300 python3 scripts/ragged_bench.py 2>&1 | grep -v amdgpu.ids | grep "441\|480"
echo "== ragged sizes: now"; timeout -k 10 300 python3 scripts/ragged_bench.py 2>&1 | grep -v amdgpu.ids | grep "441\|480"
done 2>&1 | tee gpurun_out/r5c/ragged_cr_ab.txt
