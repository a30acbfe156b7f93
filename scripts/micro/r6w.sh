cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6w; mkdir -p $O; cd $R
for w in config2 config3 config4; do
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVES --output-format csv -d $O/$w -o p -- python3 bench.py --workload $w --steps 20 --warmup 16 --no-cpu-baseline --host-io 0 --no-kernel-timing > $O/$w.log 2>&1
  echo "$w under --pmc: exit $?" | tee -a $O/summary.txt
  grep -h "gave up\|Error\|\"metric\"" $O/$w.log | cut -c1-200 | tee -a $O/summary.txt
  rm -rf $O/$w
done
timeout -k 10 300 python -m pytest tests/test_gpu_chained.py -x -q -k "under_a_tool" 2>&1 | tail -3
bash scripts/pmc_configs.sh r6w/pmc_configs > $O/pmc_configs.log 2>&1; tail -3 $O/pmc_configs.log
