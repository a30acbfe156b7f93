mkdir -p gpurun_out/r5e
for rep in 1 2; do
for f in 0 0x10000; do
echo "== OALSFX_DEBUG_FLAGS=$f (0x10000: three streams in turn for the uniform workload too)"
OALSFX_DEBUG_FLAGS=$f timeout -k 10 200 python3 scripts/chain_probe.py 4096 256 400 2>&1 | grep -v amdgpu.ids | tail -2
OALSFX_DEBUG_FLAGS=$f timeout -k 10 200 python3 scripts/chain_probe.py 4096 256 20 2>&1 | grep -v amdgpu.ids | tail -2
done; done 2>&1 | tee gpurun_out/r5e/chain_depth_uniform.txt
