mkdir -p gpurun_out/r4e
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4e/tests.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/r4e/tests.log
for rep in 1 2; do
echo "== misaligned, product"; timeout -k 10 300 python3 scripts/misaligned_bench.py 0 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4e/misaligned.txt
echo "== misaligned, 0x4000 (as round 3)"; OALSFX_DEBUG_FLAGS=0x4000 timeout -k 10 300 python3 scripts/misaligned_bench.py 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4e/misaligned.txt
done
