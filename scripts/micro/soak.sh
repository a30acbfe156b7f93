# fuzz soaks on the round's final build: bash scripts/micro/soak.sh <name>   (the logs grow as the runs go: gpurun sees progress)
mkdir -p gpurun_out/$1
OALSFX_CHAIN_FUZZ_SEEDS=${CHAIN_SEEDS:-1500} OALSFX_CHAIN_FUZZ_FULL_SIZE=${CHAIN_FULL:-40} timeout -k 10 ${LIMIT:-500} python -m pytest tests/test_gpu_chained.py -q -m gpu -k "random_runs" -x > gpurun_out/$1/chained_fuzz_soak.log 2>&1
echo "chained soak: $(tail -1 gpurun_out/$1/chained_fuzz_soak.log)"
OALSFX_FUZZ_SEEDS=${SEEDS:-1500} OALSFX_FUZZ_BATCHES=${BATCHES:-400} timeout -k 10 ${LIMIT:-500} python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random" -x > gpurun_out/$1/fuzz_soak.log 2>&1
echo "parity soak: $(tail -1 gpurun_out/$1/fuzz_soak.log)"
