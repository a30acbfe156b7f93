# What uncached delay lines / state cost or gain by themselves (chained launches): every workload of bench.py with
# OALSFX_RING_MEMORY=default and =uncached (forced for every batch), calls in plain stream order both times (--no-chain), then the
# multichannel and call-size sweeps.  Through gpurun: bash scripts/uncached_memory_bench.sh > gpurun_out/<dir>/uncached_memory.txt
for m in default uncached; do
  echo "== delay lines, state and hot records in $m memory (no chained launches)"
  for w in "" "--preset-mix" "--workload config3" "--workload config4" "--workload config5"; do
    OALSFX_RING_MEMORY=$m timeout -k 10 200 python3 bench.py $w --no-chain --no-cpu-baseline --host-io 0 --steps 200 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-62s %10.1f Msamples/s  step %7.2f us  kernel %s us' % (d['config']['workload'][:62], d['value'], d['ms_per_step'] * 1e3, d['roofline'].get('kernel_us')))"
  done
  OALSFX_RING_MEMORY=$m timeout -k 10 200 python3 scripts/multichannel_bench.py 2>/dev/null | grep channels
  OALSFX_RING_MEMORY=$m timeout -k 10 200 python3 scripts/call_size_bench.py 2>/dev/null | grep frames | tail -4
done
