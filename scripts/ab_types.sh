# same-box A/B of two builds on the per-type bench: bash scripts/ab_types.sh <old.so>   (run through gpurun)
OLD=$1
for which in old new old new; do
  if [ $which = old ]; then export OALSFX_LIB=$PWD/$OLD; else unset OALSFX_LIB; fi
  echo "== $which"; timeout -k 10 200 python scripts/per_type_bench.py 2>/dev/null | grep step
done
