# Timeline of chained launches: rocprofv3 --kernel-trace of scripts/chain_probe.py (through gpurun): bash scripts/chain_trace.sh <out-dir-under-gpurun_out> [instances frames calls]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift
mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python3 scripts/chain_probe.py "$@" > $O/probe.log 2>&1
cat $O/probe.log | grep step
O=$O python3 - <<'PY'
import csv, glob, os
O = os.environ["O"]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "steady" in r["Kernel_Name"] or "gate" in r["Kernel_Name"] or "wave_effects" in r["Kernel_Name"] or "slot_mixed" in r["Kernel_Name"]]
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"])
with open(O + "/timeline.txt", "w") as out:
    out.write("last 40 launches: start (us), end (us), duration (us), start - previous start, end - previous end, queue, kernel\n")
    ps = pe = None
    for r in tail:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        out.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f} {'' if ps is None else f'{s - ps:7.1f}':>7} {'' if pe is None else f'{e - pe:7.1f}':>7}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:60]}\n")
        ps, pe = s, e
    ends = [int(r["End_Timestamp"]) for r in rows if "steady" in r["Kernel_Name"] or "slot_mixed" in r["Kernel_Name"]][-200:]
    out.write(f"end-to-end interval over the last {len(ends)} reverb launches: {(ends[-1] - ends[0]) / 1e3 / (len(ends) - 1):.2f} us\n")
print(open(O + "/timeline.txt").read())
os.remove(f)
PY
