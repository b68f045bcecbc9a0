#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lens_r04; mkdir -p "$OUT"
cd $ROOT
RTUS_LIB=$ROOT/variants/librtus_count.so timeout -k 10 120 python3 scripts/exp_lens_tonly.py 2>&1 | grep triples
for i in 1 2 3; do
  RTUS_LIB=$ROOT/variants/librtus_prev.so timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
  timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
done | tee $OUT/ab.txt
