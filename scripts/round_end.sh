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
