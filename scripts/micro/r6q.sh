# the headline workload by batch size: microseconds per step and per round of the chip (1024 workgroups)
mkdir -p gpurun_out/r6q
for n in 4096 8192 12288 16384 24576 32768; do
timeout -k 10 400 python bench.py --instances $n --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('reverbs x $n', d['ms_per_step'], 'ms per step,', round(d['ms_per_step']*1000/($n/4096),2), 'us per round,', d['value'])"
done | tee gpurun_out/r6q/headline_by_size.txt
