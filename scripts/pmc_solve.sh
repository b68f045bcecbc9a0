#!/bin/bash
# SQ counters of the solve pipeline's two kernels (grid trace with bracket emission, refine) at sweep and scale size.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=${1:-compat}
OUT=$ROOT/gpurun_out/pmc_solve; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for size in sweep scale; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM \
     -d $OUT/$size -o sq --output-format csv -- python3 $ROOT/scripts/run_solve_once.py $size $MODE 4 > $OUT/$size.log 2>&1 || { tail -5 $OUT/$size.log; exit 1; }
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 \
     -d $OUT/${size}_b -o sq --output-format csv -- python3 $ROOT/scripts/run_solve_once.py $size $MODE 4 > $OUT/${size}_b.log 2>&1 || true
done
python3 - <<PY
import csv, glob, os, collections
for d in sorted(glob.glob("$OUT/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rtus_s" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(d.rstrip("/")), k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
