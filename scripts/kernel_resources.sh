# Registers, spills, scratch and LDS of every kernel in a .hip source, read from the gfx950 code object's metadata:
#   bash scripts/kernel_resources.sh [oalsfxpp_amd/csrc/hip/reverb.hip]      (no GPU needed)
SRC=${1:-oalsfxpp_amd/csrc/hip/reverb.hip}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=off -fno-slp-vectorize -I$R/include -I$R/oalsfxpp_amd/csrc/host -I$R/oalsfxpp_amd/csrc/hip \
    --offload-arch=gfx950 -x hip --cuda-device-only -S $R/$SRC -o $T/k.s 2>/dev/null || exit 1
python3 - $T/k.s <<'PY'
import re, subprocess, sys
s = open(sys.argv[1]).read()
print(f"{'vgpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6}  kernel")
for b in s.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', b) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('oalsfx_hip::', '').replace('void ', '')
    print(f"{g('vgpr_count'):>5} {g('sgpr_count'):>5} {g('vgpr_spill_count'):>6} {g('sgpr_spill_count'):>6} {g('private_segment_fixed_size'):>7} {g('group_segment_fixed_size'):>6}  {name}")
PY
rm -rf $T
