# chorus / flanger: the next tile's ring values requested one tile ahead (b) against the build without (a), same box; then configs 3 and 4
mkdir -p gpurun_out/r6r
bash scripts/ab_type_libs.sh ab/liboalsfx_hip_mda0.so oalsfxpp_amd/csrc/liboalsfx_hip.so CHORUS FLANGER ECHO 2>/dev/null | tee gpurun_out/r6r/ab_moddelay_ahead.txt
for w in config3 config4; do for rep in 1 2; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6r/configs.txt
timeout -k 10 600 python -m pytest tests/ -x -q -m gpu -k "wave or chorus or flanger or light or types or config or golden" > gpurun_out/r6r/tests.log 2>&1; echo "exit $?" >> gpurun_out/r6r/tests.log; tail -4 gpurun_out/r6r/tests.log
