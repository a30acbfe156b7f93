# The driver's bench command (20 steps, 5 warm-up) with libraries given as arguments, alternating, three rounds:
#   bash scripts/bench20_ab.sh ab/liboalsfx_hip_d2.so oalsfxpp_amd/csrc/liboalsfx_hip.so
for i in 1 2 3; do
  for l in "$@"; do
    OALSFX_LIB=$l python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 2>/dev/null | tail -1 | L=$l python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('%-40s value %9.1f  step %6.2f us  kernel %6.2f us  chained %d' % (os.environ['L'], d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us'], d['roofline']['chained_launches']['calls_chained_in_timed_region']))"
  done
done
