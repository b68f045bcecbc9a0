#!/bin/bash
# SQ counters of the reference-compatible forward trace for several builds of librtus.so (RTUS_LIB picks the build):
#   gpurun -- 'bash scripts/pmc_shoot.sh scripts/librtus_r01.so ray-tracing-ultrasound_amd/librtus.so ...'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_shoot; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  RTUS_LIB=$ROOT/$lib rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR \
     -d $OUT/$tag -o sq --output-format csv -- python3 $ROOT/scripts/run_shoot_once.py 0 > $OUT/$tag.log 2>&1
  RTUS_LIB=$ROOT/$lib rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM \
     -d $OUT/${tag}_b -o sq --output-format csv -- python3 $ROOT/scripts/run_shoot_once.py 0 > $OUT/${tag}_b.log 2>&1 || true
done
python3 - <<PY
import csv, glob, os, collections
for d in sorted(glob.glob("$OUT/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rtus_shoot_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc: continue
    w = None
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    print(os.path.basename(d.rstrip("/")), {k: round(v) for k, v in m.items()})
PY
