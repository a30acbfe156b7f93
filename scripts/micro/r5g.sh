mkdir -p gpurun_out/r5g
for rep in 1 2; do for st in 0 6 10 14; do
OALSFX_CHAIN_STAGGER_US=$st timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stagger $st us: driver-20', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r5g/stagger.txt
for st in 0 10; do OALSFX_CHAIN_STAGGER_US=$st timeout -k 10 300 python bench.py --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stagger $st us: default-200', d['ms_per_step'], d['value'])"; done | tee -a gpurun_out/r5g/stagger.txt
