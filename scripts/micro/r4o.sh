mkdir -p gpurun_out/r4o
for w in config3 config4; do
  for lib in ab/liboalsfx_hip_r03.so ""; do
    if [ -n "$lib" ]; then export OALSFX_LIB=$PWD/$lib; tag=r03; else unset OALSFX_LIB; tag=now; fi
    for rep in 1 2; do
      timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w $tag', d['ms_per_step'], d['value'])"
    done
  done
done 2>&1 | tee gpurun_out/r4o/config34_lfo_hoist.txt
