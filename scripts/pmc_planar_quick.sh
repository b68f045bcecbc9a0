#!/bin/bash
# SQ counters of the headline planar kernel only (two passes), printed per (element, target) solve:  gpurun -- 'bash scripts/pmc_planar_quick.sh [tag]'
set -e
TAG=${1:-quick}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_planar_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
B=$ROOT/bench.py
PMC="--steps 5 --warmup 2 --graph off --no-extra --no-cpu-baseline"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
rocprofv3 --pmc $SQ -d $OUT/sq -o sq --output-format csv -- python3 $B $PMC > $OUT/sq.log 2>&1
rocprofv3 --pmc $SQ2 -d $OUT/sq2 -o sq --output-format csv -- python3 $B $PMC > $OUT/sq2.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for d in ("sq", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rtus_tt_layers_kernel" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
solves = 64 * 1048576 / 64          # wave-solves per launch (configs[2]: 64 elements x 1,048,576 targets)
for k in sorted(tot):
    v = tot[k] / n[k]
    print(f"{k:28s} per launch {v:14.0f}   per wave-solve {v / solves:8.2f}")
PY
