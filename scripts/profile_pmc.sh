# usage: bash scripts/profile_pmc.sh <outdir-under-gpurun_out>   (run through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export OALSFX_TRAFFIC_REFRESH=1   # bench.py must not insist on a traffic.json of this build: these passes produce it
# calls in plain stream order (--no-chain): a launch's duration is then the kernel's own, which is what bench.py's roofline object reports
# (its event-timed region runs one launch after the other as well); the product's default, consecutive calls overlapping, is traced
# beside it -- there a launch's duration includes its wait for the launch before, and two launches are in flight at a time
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 50 --warmup 64 --no-cpu-baseline --no-chain > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chained -o t -- python3 bench.py --steps 50 --warmup 64 --no-cpu-baseline --no-kernel-timing > $O/bench_trace_chained.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -o p -- python3 bench.py --steps 10 --warmup 64 --no-cpu-baseline --no-chain > $O/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -o p -- python3 bench.py --steps 10 --warmup 64 --no-cpu-baseline --no-chain > $O/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --steps 10 --warmup 64 --no-cpu-baseline --no-chain > $O/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py --steps 10 --warmup 64 --no-cpu-baseline --no-chain > $O/bench_pmc4.log 2>&1
O=$O python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["O"]
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row["Kernel_Name"][:70], row["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    with open(f.replace(".csv","_summary.txt"),"w") as out:
        for (kn,cn),(n,v) in sorted(agg.items()):
            out.write(f"{kn:70s} {cn:28s} dispatches={n:5d} mean={v/n:.6g}\n")
    os.remove(f)
for f in glob.glob(O+"/**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
cat $O/trace/t_kernel_stats.csv
cat $O/pmc_*/*summary.txt | grep -i reverb
