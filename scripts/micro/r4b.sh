mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py -x -q -m gpu > gpurun_out/r4b/tests.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/r4b/tests.log
echo "== misaligned, product"; timeout -k 10 300 python3 scripts/misaligned_bench.py 0 16 37 100 441 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4b/misaligned.txt
echo "== misaligned, without the line-aligned build (0x4000)"; OALSFX_DEBUG_FLAGS=0x4000 timeout -k 10 300 python3 scripts/misaligned_bench.py 0 100 441 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4b/misaligned.txt
echo "== A/B headline: r03 vs product"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_r03.so oalsfxpp_amd/csrc/liboalsfx_hip.so 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r4b/ab_feed.txt
echo "== A/B headline: crf0 vs product"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_crf0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/r4b/ab_feed.txt
echo "== A/B wall (chained): crf0 vs product"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_crf0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 eax 256 --wall 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/r4b/ab_feed.txt
