mkdir -p gpurun_out/r4r
for w in type:CHORUS type:FLANGER; do
echo "== $w: r03 vs now"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_r03.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 $w 2>&1 | grep "per batch\|b / a"
done 2>&1 | tee gpurun_out/r4r/ab_types.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "chorus or flanger or random or golden or every" 2>&1 | tail -3
