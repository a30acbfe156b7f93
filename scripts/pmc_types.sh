# VALU counters of the ring-light kernel per effect type: bash scripts/pmc_types.sh   (run through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_types; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $O -o p -- python3 scripts/per_type_bench.py > $O/run.log 2>&1
python3 - <<'PY'
import csv, collections, glob, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_types"
f=glob.glob(O+"/**/*counter_collection.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_wave_effects" in r["Kernel_Name"] or "k_reverb_steady" in r["Kernel_Name"]]
# dispatches in order; per_type_bench runs the types in ascending order, 216 launches each
by_disp=collections.OrderedDict()
for r in rows:
    by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]]=float(r["Counter_Value"])
disp=sorted(by_disp)
names=["null","chorus","compressor","dedicated_dialog","dedicated_lfe","distortion","echo","equalizer","flanger","ring_modulator","reverb","eax_reverb"]
per=len(disp)//len(names)
with open(O+"/per_type_counters.txt","w") as out:
    out.write(f"{'type':18s} {'VALU insts/wave':>16s} {'LDS insts/wave':>15s} {'VALU active / wave cycles':>26s}\n")
    for k,nm in enumerate(names):
        d=[by_disp[i] for i in disp[k*per+per//2:(k+1)*per]]
        avg=lambda c: sum(x.get(c,0) for x in d)/len(d)
        w=avg("SQ_WAVES")
        out.write(f"{nm:18s} {avg('SQ_INSTS_VALU')/w:16.0f} {avg('SQ_INSTS_LDS')/w:15.0f} {avg('SQ_ACTIVE_INST_VALU')/max(avg('SQ_WAVE_CYCLES'),1):26.3f}\n")
os.remove(f)
print(open(O+"/per_type_counters.txt").read())
PY
