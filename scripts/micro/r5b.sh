mkdir -p gpurun_out/r5b
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5b/tests.log
for rep in 1 2; do
echo "== ragged sizes: the commit before (c_prev)"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_c_prev.so timeout -k 10 300 python3 scripts/ragged_bench.py 2>&1 | grep -v amdgpu.ids
echo "== ragged sizes: now"; timeout -k 10 300 python3 scripts/ragged_bench.py 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r5b/ragged_ab.txt
