mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4d/tests.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/r4d/tests.log
for rep in 1 2; do
echo "== misaligned, product"; timeout -k 10 300 python3 scripts/misaligned_bench.py 0 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4d/misaligned.txt
echo "== misaligned, no partial line at the call's end (ablation)"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_crabl.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 37 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4d/misaligned.txt
echo "== crf0 (no feed carry)"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_crf0.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4d/misaligned.txt
echo "== sa128"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_sa128.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4d/misaligned.txt
done
echo "== A/B kernel: crf0 vs product"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_crf0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 2>&1 | grep -v amdgpu.ids | grep "per batch\|b / a" | tee gpurun_out/r4d/ab_feed.txt
echo "== A/B wall (chained): crf0 vs product"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_crf0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 eax 256 --wall 2>&1 | grep -v amdgpu.ids | grep "per batch\|b / a" | tee -a gpurun_out/r4d/ab_feed.txt
