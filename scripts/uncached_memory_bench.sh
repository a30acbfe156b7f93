# What uncached delay lines / state cost or gain (chained launches): every workload of bench.py with OALSFX_RING_MEMORY=default
# and =uncached (forced for every batch), the multichannel and call-size sweeps.  Through gpurun: bash scripts/uncached_memory_bench.sh
for m in default uncached; do
  echo "== rings in $m memory"
  for w in "--preset-mix" "--workload config3" "--workload config4" "--workload config5"; do
    OALSFX_RING_MEMORY=$m timeout -k 10 200 python3 bench.py $w --no-cpu-baseline --host-io 0 --steps 100 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d['ms_per_step'])"
  done
  OALSFX_RING_MEMORY=$m python3 scripts/multichannel_bench.py 2>/dev/null | grep channels
  OALSFX_RING_MEMORY=$m python3 scripts/call_size_bench.py 2>/dev/null | grep frames | tail -4
done
