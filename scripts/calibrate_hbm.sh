# FETCH_SIZE / WRITE_SIZE calibration for dword-per-lane streaming (run through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
cat > $O/sweep.py <<'PY'
import sys
sys.path.insert(0, ".")
from oalsfxpp_amd import lib
so = lib.load()
write = int(sys.argv[1])
assert so.oalsfx_debug_hbm_sweep(0, 2 << 30, write, 4)
PY
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_rd -o p -- python3 $O/sweep.py 0 > $O/l1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_wr -o p -- python3 $O/sweep.py 1 > $O/l2.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sweep" in row["Kernel_Name"]:
            print(f.split("/")[-3], row["Kernel_Name"][:30], row["Counter_Name"], row["Counter_Value"])
PY
