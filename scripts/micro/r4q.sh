mkdir -p gpurun_out/r4q
for c in c_0cce45d c_696a46e c_5796c84; do
echo "== eax: r03 vs $c"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_r03.so ab/liboalsfx_hip_$c.so 4096 eax 2>&1 | grep "per batch\|b / a"
done 2>&1 | tee gpurun_out/r4q/bisect.txt
echo "== eax: r03 vs now"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_r03.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 eax 2>&1 | grep "per batch\|b / a" | tee -a gpurun_out/r4q/bisect.txt
echo "== chorus: r03 vs c_5796c84 (before the LFO change)"; timeout -k 10 300 python3 scripts/ab_libs.py ab/liboalsfx_hip_r03.so ab/liboalsfx_hip_c_5796c84.so 4096 type:CHORUS 2>&1 | grep "per batch\|b / a" | tee -a gpurun_out/r4q/bisect.txt
