mkdir -p gpurun_out/r4c
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py -x -q -m gpu > gpurun_out/r4c/tests.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/r4c/tests.log
for rep in 1 2; do
echo "== misaligned, product"; timeout -k 10 300 python3 scripts/misaligned_bench.py 0 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c/misaligned.txt
echo "== misaligned, no partial lines at the call's ends (ablation)"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_crabl.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c/misaligned.txt
echo "== misaligned, stores rounded (sa128)"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_sa128.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c/misaligned.txt
done
