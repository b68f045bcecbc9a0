#!/bin/bash
# SQ counters of the headline kernel only (one pass of scripts/profile_round.sh): gpurun -- 'bash scripts/pmc_sq_cfg2.sh'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_sq; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d $OUT/sq -o sq -- python3 $ROOT/bench.py --steps 20 --warmup 5 --graph off --no-extra --no-cpu-baseline > $OUT/sq.log 2>&1
python3 - <<PY
import sqlite3, glob
c = sqlite3.connect(glob.glob("$OUT/sq/*.db")[0])
rows = c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%rtus_tt_layers_kernel<2, false>%' group by counter_name").fetchall()
d = dict(rows); w = d["SQ_WAVES"]
print({k: round(v / w, 1) for k, v in d.items()})
PY
