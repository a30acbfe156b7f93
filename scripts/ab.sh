# same-box A/B of two builds of the library: bash scripts/ab.sh <old.so> [bench args]   (run through gpurun)
OLD=$1; shift
for round in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export OALSFX_LIB=$PWD/$OLD; else unset OALSFX_LIB; fi
    timeout -k 10 120 python bench.py --steps 200 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$which', d['roofline']['kernel_us'], d['ms_per_step'])"
  done
done
