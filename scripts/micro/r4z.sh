mkdir -p gpurun_out/r4z
timeout -k 10 900 python -m pytest tests/test_gpu_proven.py tests/test_gpu_chained.py tests/test_gpu_filters_inside.py -x -q -m gpu > gpurun_out/r4z/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4z/tests.log
for rep in 1 2 3; do
echo "== kernel by events: hand-back behind the loop (a) vs in front of the last S5 (b)"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_ehb0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 2>&1 | grep "per batch\|b / a"
echo "== whole steps, chained"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_ehb0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 eax 256 --wall 2>&1 | grep "per batch\|b / a"
done 2>&1 | tee gpurun_out/r4z/early_handback_ab.txt
