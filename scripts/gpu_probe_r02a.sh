#!/bin/bash
# round-2 first probe: instruction issue costs, counter list, SQ counters of the cfg3 planar launch, steps-20 vs steps-2000
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r02a; rm -rf "$OUT"; mkdir -p "$OUT"
$ROOT/scripts/ubench_issue > $OUT/ubench_issue.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
B=$ROOT/bench.py
S3="--workload cfg3_planar --steps 5 --warmup 2 --graph off --no-extra --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d $OUT/sq3 -o sq --output-format csv -- python3 $B $S3 > $OUT/sq3.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE \
  -d $OUT/sq3b -o sq --output-format csv -- python3 $B $S3 > $OUT/sq3b.log 2>&1 || true
python3 $B --workload cfg3_planar --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/b20.json 2> $OUT/b20.err
python3 $B --workload cfg3_planar --steps 2000 --warmup 5 --no-extra --no-cpu-baseline > $OUT/b2000.json 2> $OUT/b2000.err
cat $OUT/b20.json $OUT/b2000.json
