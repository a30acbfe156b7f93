# same-box A/B of two builds on a bench workload: bash scripts/ab_cfg.sh <old.so> [bench args]   (run through gpurun)
OLD=$1; shift
for which in old new old new old new; do
  if [ $which = old ]; then export OALSFX_LIB=$PWD/$OLD; else unset OALSFX_LIB; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$which', d['ms_per_step'], d['value'])"
done
