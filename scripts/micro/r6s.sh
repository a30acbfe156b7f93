mkdir -p gpurun_out/r6s
bash scripts/ab_type_libs.sh ab/liboalsfx_hip_mda0.so oalsfxpp_amd/csrc/liboalsfx_hip.so CHORUS FLANGER 2>/dev/null | tee gpurun_out/r6s/ab_moddelay_ahead.txt
timeout -k 10 300 python3 scripts/chorus_delay_bench.py 2>/dev/null | grep us | tee gpurun_out/r6s/chorus_delays.txt
for w in config3 config4; do for rep in 1 2; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6s/configs.txt
bash scripts/micro/suite.sh r6s
