set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 100 --warmup 64 --no-cpu-baseline ${2:-} > $O/bench_trace.log 2>&1
cat $O/trace/t_kernel_stats.csv; rm -f $O/trace/*kernel_trace.csv
