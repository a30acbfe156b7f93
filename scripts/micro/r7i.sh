# a tool the library does not recognise (rocprofv3 --pmc with the recognition switched off): the first gate gives up after its full wait, the ones behind it do not wait, the results are whole, the batch goes on in stream order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r7i; mkdir -p $O; cd $R
for w in config2 config3; do
  ( time OALSFX_IGNORE_TOOLS=1 timeout -k 10 400 rocprofv3 --pmc SQ_WAVES --output-format csv -d $O/$w -o p -- python3 bench.py --workload $w --steps 20 --warmup 16 --no-cpu-baseline --host-io 0 --no-kernel-timing --no-other-configs > $O/$w.log 2>&1 ) 2> $O/$w.time
  echo "$w under --pmc, unrecognised: exit $? $(grep real $O/$w.time)" | tee -a $O/summary.txt
  grep -h "oalsfx:\|gave up\|Error\|\"metric\"" $O/$w.log | cut -c1-260 | tee -a $O/summary.txt
  rm -rf $O/$w
done
