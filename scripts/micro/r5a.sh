mkdir -p gpurun_out/r5a
for rep in 1 2 3; do
echo "== whole steps, chained: the word stored relaxed behind s_waitcnt (a) vs as an agent-scope release store (b)"; timeout -k 10 300 python3 scripts/ab_libs.py oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_rel.so 4096 eax 256 --wall 2>&1 | grep "per batch\|b / a"
done 2>&1 | tee gpurun_out/r5a/release_store_ab.txt
echo "== the chained tests on the release-store build"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_rel.so timeout -k 10 600 python -m pytest tests/test_gpu_chained.py -q -m gpu 2>&1 | tail -2 | tee -a gpurun_out/r5a/release_store_ab.txt
bash scripts/micro/suite.sh r5a
CHAIN_SEEDS=3000 CHAIN_FULL=60 SEEDS=3000 BATCHES=600 LIMIT=400 bash scripts/micro/soak.sh r5a
