# End-of-round evidence in one gpurun call: bash scripts/round_end.sh <name>  -> gpurun_out/<name>/...
set -e
N=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$N; mkdir -p $O
bash scripts/profile_pmc.sh $N > $O/profile_pmc.log 2>&1
echo "profiles done" >> $O/progress.txt
cd $R
python3 bench.py --host-io 20 > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done" >> $O/progress.txt
for w in "--preset-mix" "--workload config3" "--workload config4" "--workload config5"; do
  python3 bench.py $w --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_other_workloads.json
done
echo "other benches done" >> $O/progress.txt
python3 scripts/per_type_bench.py 2>/dev/null | grep step > $O/per_effect_type.txt
python3 scripts/call_size_bench.py 2>/dev/null | grep frames > $O/call_sizes.txt
python3 scripts/ragged_bench.py 2>/dev/null | grep frames > $O/ragged_call_sizes.txt
python3 scripts/multichannel_bench.py 2>/dev/null | grep -v "^$" > $O/multichannel_reverb.txt || true
python3 scripts/send_filter_bench.py 2>/dev/null | grep -v "^$" > $O/send_filters.txt || true
python3 scripts/update_storm_bench.py 2>/dev/null | grep updates > $O/update_storm.txt
echo "all done" >> $O/progress.txt
cat $O/trace/t_kernel_stats.csv | cut -c1-200
# round 2 additions
python3 scripts/host_io_bench.py 2>/dev/null | grep oalsfx > $O/host_io.txt || true
python3 scripts/config4_parts.py 2>/dev/null | grep instances > $O/config4_parts.txt || true
python3 scripts/chorus_delay_bench.py 2>/dev/null | grep us > $O/chorus_delays.txt || true
if [ -f ab/liboalsfx_hip_r01.so ]; then
  python3 scripts/ab_libs.py ab/liboalsfx_hip_r01.so oalsfxpp_amd/csrc/liboalsfx_hip.so 2>/dev/null | grep -v amdgpu > $O/ab_round1_vs_round2.txt || true
  python3 scripts/ab_libs.py ab/liboalsfx_hip_r01.so oalsfxpp_amd/csrc/liboalsfx_hip.so 4096 presets 2>/dev/null | grep -v amdgpu > $O/ab_round1_vs_round2_presets.txt || true
fi
OALSFX_TRAFFIC_REFRESH=1 OALSFX_DEBUG_TIMELINE=/tmp/tl.bin python3 bench.py --steps 20 --warmup 64 --no-cpu-baseline --host-io 0 > $O/bench_timeline_build.json 2>/dev/null || true
K=$(python3 -c "import json; print([json.loads(l) for l in open('$O/bench_timeline_build.json') if l.startswith('{')][-1]['roofline']['kernel_us'])")
python3 scripts/timeline.py /tmp/tl.bin $K > $O/timeline_steady_kernel.txt || true
echo "round 2 additions done" >> $O/progress.txt
# later in round 2: micro-benchmarks behind DESIGN 3.2, and the ring-light types against round 1's library
hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 scripts/micro/chain_step.hip -o /tmp/chain_step 2>/dev/null && timeout -k 10 120 /tmp/chain_step > $O/chain_step.txt 2>&1 || true
hipcc -O3 --offload-arch=gfx950 scripts/micro/placement.hip -o /tmp/placement 2>/dev/null && timeout -k 10 60 /tmp/placement 1024 22272 > $O/workgroup_placement.txt 2>&1 || true
if [ -f ab/liboalsfx_hip_r01.so ]; then
  bash scripts/ab_type_libs.sh ab/liboalsfx_hip_r01.so oalsfxpp_amd/csrc/liboalsfx_hip.so 2>/dev/null | grep -E "==|median|b / a" > $O/ab_round1_vs_round2_ring_light_types.txt || true
fi
echo "micro-benchmarks done" >> $O/progress.txt
# round 3 additions
python3 scripts/kinds_presets_bench.py 2>/dev/null | grep step > $O/kinds_presets.txt || true
python3 scripts/low_rate_bench.py 2>/dev/null | grep step > $O/low_rates.txt || true
OALSFX_DEBUG_FLAGS=0x200 python3 scripts/send_filter_bench.py 2>/dev/null | grep -v "^$" > $O/send_filters_with_the_pre_pass_kernel.txt || true
OALSFX_DEBUG_FLAGS=0x40000000 python3 scripts/update_storm_bench.py 2>/dev/null | grep updates > $O/update_storm_without_the_cross_fading_build.txt || true
python3 scripts/overlap_probe.py 2>/dev/null | grep step > $O/overlap_probe.txt || true
bash scripts/storm_kernels.sh $N/storm4 4 > /dev/null 2>&1 || true
bash scripts/gap_trace.sh $N/gap > /dev/null 2>&1 || true
bash scripts/pmc_configs.sh $N/pmc_configs > /dev/null 2>&1 || true
bash scripts/pmc_types.sh > /dev/null 2>&1 && cp $R/gpurun_out/pmc_types/per_type_counters.txt $O/per_effect_type_counters.txt || true
bash scripts/pmc_mem.sh $N/pmc_mem > /dev/null 2>&1 || true
echo "round 3 additions done" >> $O/progress.txt
# late round 3: chained launches
bash scripts/chained_evidence.sh $N/chained > /dev/null 2>&1 || true
bash scripts/uncached_memory_bench.sh > $O/chained/uncached_memory.txt 2>&1 || true
bash scripts/storm_kernels.sh $N/chained/storm4 4 > /dev/null 2>&1 && cp $O/chained/storm4/storm_kernel_timeline.txt $O/chained/storm4_chained_timeline.txt || true
OALSFX_DEBUG_FLAGS=0x400 python3 scripts/update_storm_bench.py 2>/dev/null | grep updates > $O/update_storm_in_stream_order.txt || true
python3 bench.py --no-chain --no-cpu-baseline --host-io 0 2>/dev/null | tail -1 > $O/bench_default_no_chain.json || true
echo "chained launches done" >> $O/progress.txt
