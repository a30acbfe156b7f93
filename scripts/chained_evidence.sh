# Evidence behind DESIGN 4 "chained launches" in one gpurun call: bash scripts/chained_evidence.sh <name>  -> gpurun_out/<name>/...
N=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N; mkdir -p $O
cd $R
bash scripts/bench20.sh > $O/bench_steps20_warmup5.txt 2>/dev/null
bash scripts/bench20_spin.sh > $O/bench20_spin_up.txt 2>/dev/null
bash scripts/chain_trace.sh $N/trace_chained > /dev/null 2>&1 && cp $O/trace_chained/timeline.txt $O/timeline_chained.txt && cp $O/trace_chained/probe.log $O/probe_chained.txt
OALSFX_DEBUG_FLAGS=0x400 bash scripts/chain_trace.sh $N/trace_stream_order > /dev/null 2>&1 && cp $O/trace_stream_order/timeline.txt $O/timeline_stream_order.txt && cp $O/trace_stream_order/probe.log $O/probe_stream_order.txt
for f in 64 128 256 512 1024 2048; do
  for n in 1024 2048 4096; do
    echo "chained:      $(timeout -k 10 100 python3 scripts/chain_probe.py $n $f 300 2>/dev/null | tail -1)"
    echo "stream order: $(OALSFX_DEBUG_FLAGS=0x400 timeout -k 10 100 python3 scripts/chain_probe.py $n $f 300 2>/dev/null | tail -1)"
  done
done > $O/instances_and_call_sizes.txt
if [ -f ab/liboalsfx_hip_cx1.so ]; then
  echo "product (no cache invalidated behind the wait for a turn):" > $O/acquire_cost.txt
  timeout -k 10 100 python3 scripts/chain_probe.py 2>/dev/null | grep step >> $O/acquire_cost.txt
  echo "with an agent-scope acquire (buffer_inv sc1) behind the wait (-DOALSFX_CHAIN_EXP=1):" >> $O/acquire_cost.txt
  OALSFX_LIB=ab/liboalsfx_hip_cx1.so timeout -k 10 100 python3 scripts/chain_probe.py 2>/dev/null | grep step >> $O/acquire_cost.txt
  echo "stream order:" >> $O/acquire_cost.txt
  OALSFX_DEBUG_FLAGS=0x400 timeout -k 10 100 python3 scripts/chain_probe.py 2>/dev/null | grep step >> $O/acquire_cost.txt
fi
echo "chained:" > $O/send_filters.txt; timeout -k 10 200 python3 scripts/send_filter_bench.py 2>/dev/null | grep -v "^$" >> $O/send_filters.txt
echo "stream order:" >> $O/send_filters.txt; OALSFX_DEBUG_FLAGS=0x400 timeout -k 10 200 python3 scripts/send_filter_bench.py 2>/dev/null | grep -v "^$" >> $O/send_filters.txt
echo "chained:" > $O/kinds_presets.txt; timeout -k 10 200 python3 scripts/kinds_presets_bench.py 2>/dev/null | grep step >> $O/kinds_presets.txt
echo "stream order:" >> $O/kinds_presets.txt; OALSFX_DEBUG_FLAGS=0x400 timeout -k 10 200 python3 scripts/kinds_presets_bench.py 2>/dev/null | grep step >> $O/kinds_presets.txt
for d in 2; do
  if [ -f ab/liboalsfx_hip_d$d.so ]; then
    echo "launches in flight: $d" >> $O/chain_depth.txt
    OALSFX_LIB=ab/liboalsfx_hip_d$d.so timeout -k 10 100 python3 scripts/chain_probe.py 2>/dev/null | tail -1 >> $O/chain_depth.txt
    OALSFX_LIB=ab/liboalsfx_hip_d$d.so timeout -k 10 200 python3 scripts/update_storm_bench.py 4 2>/dev/null | grep updates >> $O/chain_depth.txt
    OALSFX_LIB=ab/liboalsfx_hip_d$d.so timeout -k 10 200 python3 scripts/kinds_presets_bench.py 2>/dev/null | grep "all (i" >> $O/chain_depth.txt
  fi
done
echo "launches in flight: 3 (the product)" >> $O/chain_depth.txt
timeout -k 10 100 python3 scripts/chain_probe.py 2>/dev/null | tail -1 >> $O/chain_depth.txt
timeout -k 10 200 python3 scripts/update_storm_bench.py 4 2>/dev/null | grep updates >> $O/chain_depth.txt
timeout -k 10 200 python3 scripts/kinds_presets_bench.py 2>/dev/null | grep "all (i" >> $O/chain_depth.txt
echo done > $O/progress.txt
