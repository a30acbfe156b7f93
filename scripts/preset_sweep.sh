# per-preset cost of the reverb path: bash scripts/preset_sweep.sh "<indices>" [instances]   (run through gpurun)
N=${2:-4096}
for p in $1; do
  timeout -k 10 120 python bench.py --steps 50 --warmup 64 --no-cpu-baseline --preset $p --instances $N 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$p', d['config']['workload'][-40:], d['kernels'], d['ms_per_step'])"
done
