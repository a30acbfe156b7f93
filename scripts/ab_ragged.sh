# same-box A/B of two builds on the ragged-call bench: bash scripts/ab_ragged.sh <old.so>   (run through gpurun)
OLD=$1
for which in old new old new; do
  if [ $which = old ]; then export OALSFX_LIB=$PWD/$OLD; else unset OALSFX_LIB; fi
  echo "== $which"; timeout -k 10 200 python scripts/ragged_bench.py 2>/dev/null | grep frames
done
