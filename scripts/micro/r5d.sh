mkdir -p gpurun_out/r5d
timeout -k 10 900 python -m pytest tests/test_gpu_chained.py tests/test_gpu_proven.py -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5d/tests.log; grep -E "^E  " gpurun_out/r5d/tests.log | head
for rep in 1 2 3 4; do
echo "== whole steps, chained: CU names in front of the record (a) vs beside it (b)"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_cub0.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 eax 256 --wall 2>&1 | grep "per batch\|b / a"
done 2>&1 | tee gpurun_out/r5d/cu_check_beside_record_ab.txt
