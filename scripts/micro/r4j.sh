mkdir -p gpurun_out/r4j
for rep in 1 2 3 4; do
echo "== pool off, blocks as allocated (run $rep)"; OALSFX_UNCACHED_POOL=0 timeout -k 10 300 python3 scripts/uncached_free_hazard.py 1000 1000 2>&1 | grep "bad buffers" | tr '\n' ';'; echo
echo "== pool off, every uncached block whole 2 MiB granules (run $rep)"; OALSFX_UNCACHED_GRANULE_MB=2 OALSFX_UNCACHED_POOL=0 timeout -k 10 300 python3 scripts/uncached_free_hazard.py 1000 1000 2>&1 | grep "bad buffers" | tr '\n' ';'; echo
done 2>&1 | tee gpurun_out/r4j/granule.txt
echo "== suite subset, pool off, 2 MiB granules"; OALSFX_UNCACHED_GRANULE_MB=2 OALSFX_UNCACHED_POOL=0 OALSFX_FUZZ_BATCHES=120 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r4j/granule.txt
echo "== suite subset, pool off, as allocated"; OALSFX_UNCACHED_POOL=0 OALSFX_FUZZ_BATCHES=120 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r4j/granule.txt
