# Per effect type: two builds of the library timed alternately in one process (scripts/ab_libs.py).
#   bash scripts/ab_type_libs.sh <a.so> <b.so> [types...]
A=${1:-ab/liboalsfx_hip_base.so}; B=${2:-oalsfxpp_amd/csrc/liboalsfx_hip.so}; shift; shift
for t in ${@:-NULL CHORUS ECHO DISTORTION EQUALIZER COMPRESSOR RING_MODULATOR}; do echo "== $t"; python scripts/ab_libs.py $A $B 4096 type:$t | grep -E "median|b / a"; done
