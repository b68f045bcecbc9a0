#!/bin/bash
# fuzz of the forward trace with the conditioning-aware criterion, for the working tree and (RTUS_LIB) the round-1 trace
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fuzz; mkdir -p "$OUT"
cd $ROOT
N=${1:-60}; SEED=${2:-777}
timeout -k 10 900 python3 scripts/fuzz_shoot.py $N $SEED > $OUT/new_$SEED.txt 2>&1; echo "new rc=$?"; tail -4 $OUT/new_$SEED.txt
RTUS_LIB=$ROOT/scripts/librtus_shoot_r01.so timeout -k 10 900 python3 scripts/fuzz_shoot.py $N $SEED > $OUT/r01_$SEED.txt 2>&1; echo "r01 rc=$?"; tail -4 $OUT/r01_$SEED.txt
timeout -k 10 300 python3 scripts/fuzz_shoot.py 6 4242 --selftest > $OUT/selftest.txt 2>&1; echo "selftest rc=$? (3 = caught)"; tail -2 $OUT/selftest.txt
