# does a run of chained launches survive rocprofv3 --pmc (which serialises kernels)?  headline, configs[2], configs[3]; and what the tool puts in the environment
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6v; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_WAVES --output-format csv -d $O/env -o p -- python3 -c "import os; print('\n'.join(k+'='+v[:120] for k,v in sorted(os.environ.items()) if 'ROC' in k.upper() or 'HSA' in k.upper() or 'LD_PRELOAD' in k))" > $O/env.txt 2>&1
for w in config2 config3 config4; do
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVES --output-format csv -d $O/$w -o p -- python3 bench.py --workload $w --steps 20 --warmup 16 --no-cpu-baseline --host-io 0 --no-kernel-timing > $O/$w.log 2>&1
  echo "$w under --pmc: exit $?" | tee -a $O/summary.txt
  grep -h "gave up\|Error\|\"metric\"" $O/$w.log | cut -c1-300 | tee -a $O/summary.txt
  rm -rf $O/$w
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --workload config3 --steps 20 --warmup 16 --no-cpu-baseline --host-io 0 --no-kernel-timing > $O/kt.log 2>&1
echo "config3 under --kernel-trace: exit $?" | tee -a $O/summary.txt
grep -h "gave up\|\"metric\"" $O/kt.log | cut -c1-200 | tee -a $O/summary.txt
rm -rf $O/kt
cat $O/env.txt | head -40
