#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lens_r04; mkdir -p "$OUT"
cd $ROOT
RTUS_LIB=$ROOT/variants/librtus_count.so timeout -k 10 120 python3 scripts/exp_lens_tonly.py 2>&1 | grep triples
for i in 1 2 3; do
  RTUS_LIB=$ROOT/variants/librtus_prev.so timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
  timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
done | tee $OUT/ab.txt
for i in 1 2; do
  RTUS_LIB=$ROOT/variants/librtus_prev.so timeout -k 10 120 python3 scripts/ab_lens_f64.py 2>&1 | tail -2
  timeout -k 10 120 python3 scripts/ab_lens_f64.py 2>&1 | tail -2
done | tee $OUT/ab64.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_lens_rows.py -q -m gpu -s > $OUT/rows.txt 2>&1; echo "rows rc $?" >> $OUT/rows.txt
grep -E "rows/workgroup|passed|failed|rc |Error|assert" $OUT/rows.txt | cut -c1-60,200-400
bash scripts/pmc_lens_ab.sh 2>&1 | tail -4
