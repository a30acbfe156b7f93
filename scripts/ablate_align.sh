# timing experiment: what would perfectly line-aligned tap reads be worth?  (results are wrong on purpose)
for f in 0 32 64 0; do
  echo "== OALSFX_DEBUG_FLAGS=$f"
  OALSFX_DEBUG_FLAGS=$f timeout -k 10 120 python bench.py --steps 100 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_us'], d['ms_per_step'])"
done
