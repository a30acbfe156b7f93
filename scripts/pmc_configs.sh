# VALU issue counters of every kernel of a step of BASELINE configs[2] and configs[3] (bench.py --workload config3 / config4), one rocprofv3
# --pmc pass each (no tracing alongside):  bash scripts/pmc_configs.sh <out-dir-under-gpurun_out>   (run through gpurun)
# Writes <out>/valu_issue.json: per workload the wave-level VALU instructions one step issues (all kernels of the step), which bench.py
# turns into the fraction of the chip's VALU issue slots (profiles/valu_issue.json is the committed copy).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for w in config3 config4; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/$w -o p -- python3 bench.py --workload $w --steps 20 --warmup 64 --no-cpu-baseline --host-io 0 > $O/$w.log 2>&1
done
O=$O python3 - <<'PY'
import csv, glob, json, os, collections, re
O = os.environ["O"]
out = {"note": "rocprofv3 --pmc pass of bench.py --workload <w> (scripts/pmc_configs.sh): wave-level instructions per launch, averaged over the "
               "launches of the second half of the run; a wave64 VALU instruction holds its SIMD for 4 cycles; 1024 SIMDs at 2.4 GHz"}
for w in ("config3", "config4"):
    f = glob.glob(f"{O}/{w}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "oalsfx" not in r["Kernel_Name"]: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("oalsfx_hip::", "")
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, c in per.items():
        half = lambda v: v[len(v) // 2:]
        mean = lambda name: (sum(half(c[name])) / max(len(half(c[name])), 1)) if name in c else 0.0
        if k.startswith(("k_upload", "k_fill", "k_ring_probe", "k_null")): continue
        kernels[k] = {"launches": len(c["SQ_WAVES"]), "waves": mean("SQ_WAVES"), "valu_insts": mean("SQ_INSTS_VALU"), "lds_insts": mean("SQ_INSTS_LDS"),
                      "salu_insts": mean("SQ_INSTS_SALU"), "valu_active_over_wave_cycles": round(mean("SQ_ACTIVE_INST_VALU") / max(mean("SQ_WAVE_CYCLES"), 1), 4),
                      "gui_active_cycles_per_xcd": mean("GRBM_GUI_ACTIVE") / 8}
    # the kernels of a step: launched once per step (what ran only while the batch warmed up -- the general reverb kernel of the first call -- is left out)
    most = max(v["launches"] for v in kernels.values())
    kernels = {k: v for k, v in kernels.items() if v["launches"] >= most / 2}
    out[w] = {"kernels": kernels, "valu_wave_instructions_per_step": sum(v["valu_insts"] for v in kernels.values())}
    os.remove(f)
json.dump(out, open(O + "/valu_issue.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
