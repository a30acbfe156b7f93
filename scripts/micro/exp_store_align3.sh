echo "== ordinary memory (OALSFX_RING_MEMORY=default: stream order)"; OALSFX_RING_MEMORY=default timeout -k 10 300 python3 scripts/misaligned_bench.py 0 16 100 441 0
echo "== uncached, stream order (0x400)"; OALSFX_DEBUG_FLAGS=0x400 timeout -k 10 300 python3 scripts/misaligned_bench.py 0 16 100 441 0
echo "== ragged sizes, product"; timeout -k 10 300 python3 scripts/ragged_bench.py
