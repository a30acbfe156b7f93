# timing experiments: reverb kernel with parts switched off (results are wrong on purpose)
for f in 0 1 2 4 6 7; do
  echo "== OALSFX_DEBUG_FLAGS=$f"
  OALSFX_DEBUG_FLAGS=$f timeout -k 10 120 python bench.py --steps 50 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_us'], d['ms_per_step'])"
done
