# usage: bash scripts/pmc_mem.sh <outdir-under-gpurun_out>   (run through gpurun)
# memory-pipeline counters of the reverb kernel, one rocprofv3 --pmc pass per group (no tracing alongside).
# TA_* counters are left out: a pass with them aborted inside rocprofv3 on this pool (signal 6, incomplete dispatch).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  echo "group $i: $group" >> $O/progress.txt
  timeout -k 10 120 rocprofv3 --pmc $group --output-format csv -d $O/g$i -o p -- python3 bench.py --steps 10 --warmup 64 --no-cpu-baseline > $O/g$i.log 2>&1 || { echo "group $i failed: $group" | tee -a $O/progress.txt; }
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_INFLIGHT_MAX
TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_LATENCY TCP_TCC_READ_REQ_LATENCY
TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL
TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_CACHE_ACCESSES TCP_READ_TAGCONFLICT_STALL_CYCLES
GRBM_GUI_ACTIVE TCC_BUSY TCC_CYCLE TCP_GATE_EN1
TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_UTCL1_SERIALIZATION_STALL
GROUPS
O=$O python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["O"]
for f in sorted(glob.glob(O+"/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row["Kernel_Name"][:60], row["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    with open(f.replace(".csv","_summary.txt"),"w") as out:
        for (kn,cn),(n,v) in sorted(agg.items()):
            out.write(f"{kn:60s} {cn:40s} dispatches={n:5d} mean={v/n:.6g}\n")
    os.remove(f)
PY
cat $O/g*/*summary.txt | grep steady_coop
