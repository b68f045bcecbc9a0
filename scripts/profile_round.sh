#!/bin/bash
# Re-creates every file under profiles/ for one round on the GPU box:
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r01'
# Raw rocprofv3 output lands in gpurun_out/prof_<tag>/; scripts/profile_summarise.py turns it into profiles/<tag>_*.
# rocprofv3 rules of this pool: run from /tmp with TMPDIR=/tmp, the program itself right after `--`, counters in their
# own passes (never together with a trace domain other than --kernel-trace).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
B=$ROOT/bench.py
SHORT="--steps 20 --warmup 5 --graph off --no-extra --no-cpu-baseline"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
echo "[1/7] un-profiled default bench";   python3 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "[2/7] kernel trace of the default bench"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $B > $OUT/kt.log 2>&1
echo "[3/7] WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE -d $OUT/wr -o wr -- python3 $B $SHORT > $OUT/wr.log 2>&1
echo "[4/7] FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE -d $OUT/rd -o rd -- python3 $B $SHORT > $OUT/rd.log 2>&1
echo "[5/7] SQ counters, headline kernel"; rocprofv3 --pmc $SQ -d $OUT/sq -o sq -- python3 $B $SHORT > $OUT/sq.log 2>&1
echo "[5b] WRITE_SIZE / FETCH_SIZE, configs[2] (537 MB per launch)"
SHORT3="--workload cfg3_planar --steps 5 --warmup 2 --graph off --no-extra --no-cpu-baseline"
rocprofv3 --pmc WRITE_SIZE -d $OUT/wr3 -o wr -- python3 $B $SHORT3 > $OUT/wr3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/rd3 -o rd -- python3 $B $SHORT3 > $OUT/rd3.log 2>&1
echo "[6/7] SQ counters, forward trace (reference arithmetic)"
rocprofv3 --pmc $SQ -d $OUT/sq_shoot0 -o sq -- python3 $ROOT/scripts/run_shoot_once.py 0 > $OUT/sq_shoot0.log 2>&1
echo "[7/7] SQ counters, forward trace (vector form)"
rocprofv3 --pmc $SQ -d $OUT/sq_shoot1 -o sq -- python3 $ROOT/scripts/run_shoot_once.py 1 > $OUT/sq_shoot1.log 2>&1
find $OUT -name "*.csv" | sort
