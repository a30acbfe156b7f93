for i in 1 2 3; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chained   ', d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us'], d['roofline']['chained_launches']['calls_chained_in_timed_region'])"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 --no-chain 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no chain  ', d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us'], d['roofline']['chained_launches']['calls_chained_in_timed_region'])"
done
