mkdir -p gpurun_out/r4p
for w in type:CHORUS type:FLANGER type:ECHO eax; do
echo "== $w: r03 vs now"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_r03.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 $w 2>&1 | grep "per batch\|b / a"
done 2>&1 | tee gpurun_out/r4p/ab_types.txt
