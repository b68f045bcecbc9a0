#!/bin/bash
# SQ counters of the fp32 lens kernel (configs[3] shard) for the working tree and variants/librtus_prev.so
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_lens_ab; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ3="SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for tag in prev tree; do
  if [ $tag = prev ]; then export RTUS_LIB=$ROOT/variants/librtus_prev.so; else unset RTUS_LIB; fi
  rocprofv3 --pmc $SQ -d $OUT/sq_$tag -o sq --output-format csv -- python3 $ROOT/bench.py --workload cfg4_lens_f32 --steps 5 --warmup 2 --graph off --no-extra --no-cpu-baseline > $OUT/sq_$tag.log 2>&1
  rocprofv3 --pmc $SQ3 -d $OUT/sq3_$tag -o sq --output-format csv -- python3 $ROOT/bench.py --workload cfg4_lens_f32 --steps 5 --warmup 2 --graph off --no-extra --no-cpu-baseline > $OUT/sq3_$tag.log 2>&1
done
python3 - $OUT <<'PY'
import sys,glob,csv,collections
out=sys.argv[1]
for tag in ("prev","tree"):
    for pas in ("sq","sq3"):
        agg=collections.defaultdict(lambda:[0,0])
        for f in glob.glob(f"{out}/{pas}_{tag}/**/*counter_collection.csv",recursive=True):
            for r in csv.DictReader(open(f)):
                if "lens" in r["Kernel_Name"]:
                    a=agg[r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
        print(tag,pas,{k:round(v[0]/max(v[1],1)) for k,v in agg.items()})
PY
