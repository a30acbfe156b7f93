mkdir -p gpurun_out/r4w
for rep in 1 2; do for lib in ab/liboalsfx_hip_r03.so ""; do
  if [ -n "$lib" ]; then export OALSFX_LIB=$PWD/$lib; tag=r03; else unset OALSFX_LIB; tag=now; fi
  timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config5 $tag', d['ms_per_step'], d['value'], d['roofline'].get('delay_line_placement'))"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('driver-20 $tag', d['ms_per_step'], d['value'], d['roofline']['kernel_us'])"
done; done 2>&1 | tee gpurun_out/r4w/config5_ab.txt
