# Where does an update-storm step spend its time?  rocprofv3 kernel + copy + HIP API trace of scripts/update_storm_bench.py 4 (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf $R/gpurun_out/storm_trace
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/storm_trace -o s -- python3 scripts/update_storm_bench.py 4 > $R/gpurun_out/storm_trace.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(R+"/gpurun_out/storm_trace/**/*hip_api_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows=rows[len(rows)*3//4:]          # the timed part
    agg=collections.defaultdict(lambda:[0,0])
    for r in rows:
        d=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        agg[r["Function"]][0]+=1; agg[r["Function"]][1]+=d
    tot=sum(v[1] for v in agg.values())
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
        print(f"{k:40s} calls {v[0]:6d}  total {v[1]/1000:10.1f} us  mean {v[1]/v[0]/1000:8.2f} us")
    print("sum", tot/1000)
PY
