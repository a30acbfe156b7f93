# A/B: non-temporal hints on the reverb's ring traffic (builds ab/liboalsfx_hip_nt<bits>.so, see OALSFX_NT in hip/reverb.hip)
#   bash scripts/ab_nt.sh <base: product or bits> <bits> [<bits> ...]
A=$1; shift
[ "$A" = product ] && A=oalsfxpp_amd/csrc/liboalsfx_hip.so || A=ab/liboalsfx_hip_nt$A.so
for v in "$@"; do
  echo "=== NT=$v (b) against $A (a)"
  python3 scripts/ab_libs.py $A ab/liboalsfx_hip_nt$v.so 2>/dev/null | grep -E "median|b / a"
done
