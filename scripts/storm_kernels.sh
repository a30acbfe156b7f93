# Which kernels does an update-storm step run, and when?  rocprofv3 --kernel-trace of scripts/update_storm_bench.py <k> (run through gpurun):
#   bash scripts/storm_kernels.sh <out-dir-under-gpurun_out> [updates per buffer, default 4]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; K=${2:-4}
mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o s -- python3 scripts/update_storm_bench.py $K > $O/storm.log 2>&1
O=$O python3 - <<'PY'
import csv, glob, os, re, collections
O = os.environ["O"]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void oalsfx_hip::", "", n); n = re.sub(r"\(.*", "", n)
    return n.replace("false", "f").replace("true", "T")
rows = [r for r in rows if "oalsfx" in r["Kernel_Name"]]
tail = rows[-60:]
t0 = int(tail[0]["Start_Timestamp"])
with open(O + "/storm_kernel_timeline.txt", "w") as out:
    out.write("last 60 kernels of the run: start (us), duration (us), gap to the previous kernel's end (us), kernel\n")
    prev_end = None
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:7.1f}"
        out.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:>8s}  {short(r['Kernel_Name'])}\n")
        prev_end = max(prev_end or 0, e)
    agg = collections.defaultdict(list)
    for r in rows[len(rows) // 2:]:
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out.write("\nsecond half of the run: calls, mean, median, max (us)\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        out.write(f"{len(v):6d} {sum(v) / len(v):8.1f} {v[len(v) // 2]:8.1f} {v[-1]:8.1f}  {k}\n")
os.remove(f)
print(open(O + "/storm_kernel_timeline.txt").read())
PY
grep updates $O/storm.log
