# timing experiment: what are line-aligned ring stores worth after an odd-sized call?  (results of the ablated builds are wrong on purpose)
# build the variants first (CPU side): for a in 32 64 128; do OALSFX_BUILD_TAG=sa$a OALSFX_EXTRA_FLAGS=-DOALSFX_ABLATE_STORE_ALIGN=$a python -m oalsfxpp_amd.build; done
set -e
FIRSTS="${FIRSTS:-0 8 16 24 32 37 100 441}"
echo "== product library"; timeout -k 10 300 python3 scripts/misaligned_bench.py $FIRSTS
echo "== product library, taps rounded to 128 bytes too (OALSFX_DEBUG_FLAGS=32)"; OALSFX_DEBUG_FLAGS=32 timeout -k 10 300 python3 scripts/misaligned_bench.py 0 100 441
for a in 128 64 32; do
  [ -f ab/liboalsfx_hip_sa$a.so ] || continue
  echo "== ring stores rounded down to $a bytes"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_sa$a.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 100 441
done
