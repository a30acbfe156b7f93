# VALU ablation: builds whose tile loop issues 64 / 128 / 256 more vector instructions per wavefront and tile (doing nothing), against the product: the headline chained (400 steps) and the kernel by itself (events)
mkdir -p gpurun_out/r7h
for rep in 1 2; do for lib in oalsfxpp_amd/csrc/liboalsfx_hip.so ab/liboalsfx_hip_av64.so ab/liboalsfx_hip_av128.so ab/liboalsfx_hip_av256.so; do
OALSFX_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 400 --no-cpu-baseline --host-io 0 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib: step', d['ms_per_step']*1000, 'us chained; kernel by itself', d['roofline']['kernel_us'], 'us')"
done; done | tee gpurun_out/r7h/valu_ablation.txt
