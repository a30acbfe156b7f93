# The driver's bench command with and without the device spin-up, alternating:  bash scripts/bench20_spin.sh
for i in 1 2 3; do
  for ms in 60 0 200; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 --spin-up-ms $ms 2>/dev/null | tail -1 | M=$ms python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read()); c=d['roofline']['chained_launches']; print('spin-up %4s ms: value %9.1f  step %6.2f us  kernel alone %6.2f us  steady-state interval %6.2f us' % (os.environ['M'], d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us'], c.get('steady_state_interval_us', 0)))"
  done
done
