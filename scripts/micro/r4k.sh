mkdir -p gpurun_out/r4k
echo "== today's form of the experiment: nothing waits (MAX_GIB=0), 2 MiB granules"
for rep in 1 2 3 4; do OALSFX_UNCACHED_POOL_MAX_GIB=0 timeout -k 10 300 python3 scripts/uncached_free_hazard.py 1000 1000 2>&1 | grep "bad buffers" | tr '\n' ';'; echo; done 2>&1 | tee gpurun_out/r4k/hazard_granules.txt
echo "== whole GPU suite, nothing waits for reuse (every uncached block freed when its batch goes), fuzz widened"
OALSFX_UNCACHED_POOL_MAX_GIB=0 OALSFX_FUZZ_BATCHES=120 OALSFX_FUZZ_SEEDS=60 timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r4k/suite_cap0.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r4k/suite_cap0.log
