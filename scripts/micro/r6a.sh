# config 3 (chorus, flanger, echo, EAX reverb) with the handed-on memory uncached, in stream order: what multi-slot chaining would start from
mkdir -p gpurun_out/r6a
for rep in 1 2; do for kind in default uncached; do
OALSFX_RING_MEMORY=$kind timeout -k 10 300 python bench.py --workload config3 --no-cpu-baseline --host-io 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config3 $kind', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r6a/config3_memory_kind.txt
