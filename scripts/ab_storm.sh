# same-box A/B of two builds on the update-storm bench: bash scripts/ab_storm.sh <old.so>   (run through gpurun)
OLD=$1
for which in old new old new; do
  if [ $which = old ]; then export OALSFX_LIB=$PWD/$OLD; else unset OALSFX_LIB; fi
  echo "== $which"; timeout -k 10 200 python scripts/update_storm_bench.py 0 4 40 2>/dev/null | grep updates
  timeout -k 10 100 python bench.py --instances 2048 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('2048 instances: ms_per_step', d['ms_per_step'])"
done
