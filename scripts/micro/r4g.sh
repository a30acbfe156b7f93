mkdir -p gpurun_out/r4g
for mode in 1 0 2; do for m in host device; do
echo "== OALSFX_UNCACHED_POOL=$mode MODE=$m"; OALSFX_UNCACHED_POOL=$mode MODE=$m TYPES=11,3,1,10 timeout -k 10 300 python3 scripts/uncached_free_hazard.py 1000 1000 2>&1 | grep -v amdgpu.ids | tail -14
done; done 2>&1 | tee gpurun_out/r4g/uncached_free_hazard.txt
