mkdir -p gpurun_out/r6z
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6z/bench_driver_command.json 2> gpurun_out/r6z/bench_driver_command.err ) 2> gpurun_out/r6z/time_driver.txt
( time python bench.py > gpurun_out/r6z/bench_default.json 2> gpurun_out/r6z/bench_default.err ) 2> gpurun_out/r6z/time_default.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6z/bench_driver_command_2.json 2>/dev/null
python bench.py --workload config5 --no-cpu-baseline > gpurun_out/r6z/bench_config5.json 2>/dev/null
python bench.py --config5 --no-cpu-baseline --no-other-configs --host-io 0 > gpurun_out/r6z/bench_with_config5_object.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_driver_command", "bench_default", "bench_driver_command_2", "bench_config5", "bench_with_config5_object"):
    d = [json.loads(l) for l in open(f"gpurun_out/r6z/{f}.json") if l.startswith("{")][-1]
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], {k: (v["ms_per_step"], v["value"], v.get("calls_chained")) for k, v in d.get("other_configs", {}).items()}, (d.get("config5") or {}).get("ms_per_step"))
PY
cat gpurun_out/r6z/time_driver.txt gpurun_out/r6z/time_default.txt | grep real
