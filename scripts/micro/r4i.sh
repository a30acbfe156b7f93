mkdir -p gpurun_out/r4i
for mode in 0 2; do for m in host device; do
echo "== OALSFX_UNCACHED_POOL=$mode MODE=$m"; OALSFX_UNCACHED_POOL=$mode MODE=$m timeout -k 10 300 python3 scripts/uncached_free_hazard.py 1000 1000 2>&1 | grep -v amdgpu.ids | tail -12
done; done 2>&1 | tee gpurun_out/r4i/hazard_modes.txt
echo "== traced, pool off, host path"; OALSFX_POOL_TRACE=1 OALSFX_UNCACHED_POOL=0 timeout -k 10 300 python3 scripts/uncached_free_hazard.py 1000 1000 > gpurun_out/r4i/hazard_trace.txt 2>&1; grep -c trace gpurun_out/r4i/hazard_trace.txt
