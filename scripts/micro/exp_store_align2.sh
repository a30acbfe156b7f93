echo "== product"; timeout -k 10 300 python3 scripts/misaligned_bench.py 0 16 48 80 32 64 96 0 16
for a in 64 128; do echo "== sa$a"; OALSFX_LIB=$PWD/ab/liboalsfx_hip_sa$a.so timeout -k 10 300 python3 scripts/misaligned_bench.py 0 8 16 24 37 100 441 0 16; done
