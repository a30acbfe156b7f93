# Gaps between consecutive kernels of the headline loop in plain stream order (--no-chain; chained launches overlap: scripts/chain_trace.sh): rocprofv3 --kernel-trace of bench.py (run through gpurun): bash scripts/gap_trace.sh <out-dir-under-gpurun_out> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift
mkdir -p $O; cd $R
OALSFX_TRAFFIC_REFRESH=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python3 bench.py --steps 200 --warmup 64 --no-cpu-baseline --host-io 0 --no-chain "$@" > $O/bench.log 2>&1
O=$O python3 - <<'PY'
import csv, glob, os, re
O = os.environ["O"]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "oalsfx" in r["Kernel_Name"]]
gaps, durs = [], []
for a, b in zip(rows, rows[1:]):
    if "steady" in a["Kernel_Name"] and "steady" in b["Kernel_Name"]:
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if g < 50: gaps.append(g); durs.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
# the longest run of consecutive steady-state launches: wall time per launch on the GPU's own clock
best = (0, 0, 0)
i = 0
while i < len(rows):
    j = i
    while j + 1 < len(rows) and "steady" in rows[j]["Kernel_Name"] and "steady" in rows[j + 1]["Kernel_Name"] and int(rows[j + 1]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"]) < 50000: j += 1
    if j - i > best[0]: best = (j - i, i, j)
    i = j + 1
n, i, j = best
span = (int(rows[j]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
starts = [(int(rows[k + 1]["Start_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3 for k in range(i, j)]
starts.sort()
gaps.sort(); durs.sort()
with open(O + "/gaps.txt", "w") as out:
    out.write(f"longest run: {n + 1} launches in {span:.1f} us = {span / (n + 1):.2f} us per launch; start-to-start median {starts[len(starts)//2]:.2f}, p10 {starts[len(starts)//10]:.2f}, p90 {starts[len(starts)*9//10]:.2f}\n")
    out.write(f"{len(gaps)} consecutive steady-state launches: gap end -> next start: median {gaps[len(gaps)//2]:.2f} us, p10 {gaps[len(gaps)//10]:.2f}, p90 {gaps[len(gaps)*9//10]:.2f}; kernel median {durs[len(durs)//2]:.2f} us\n")
print(open(O + "/gaps.txt").read())
os.remove(f)
PY
grep "^{" $O/bench.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench under the tracer: ms_per_step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'])"
