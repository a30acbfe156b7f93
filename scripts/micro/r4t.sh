mkdir -p gpurun_out/r4t
for f in 0 0x1000000 0 0x1000000; do echo "== OALSFX_DEBUG_FLAGS=$f"; OALSFX_DEBUG_FLAGS=$f timeout -k 10 300 python3 scripts/update_storm_bench.py 2>&1 | grep -v amdgpu.ids | tail -8; done 2>&1 | tee gpurun_out/r4t/storm_nf.txt
