#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lens_r04; mkdir -p "$OUT"
cd $ROOT
RTUS_LIB=$ROOT/variants/librtus_count_prev.so timeout -k 10 120 python3 scripts/exp_lens_tonly.py 2>&1 | grep triples
RTUS_LIB=$ROOT/variants/librtus_count.so timeout -k 10 120 python3 scripts/exp_lens_tonly.py 2>&1 | grep triples
timeout -k 10 900 python3 -m pytest tests/test_gpu_lens_rows.py -q -m gpu -s > $OUT/rows.txt 2>&1; echo "rows rc $?" >> $OUT/rows.txt
grep -E "rows/workgroup|passed|failed|rc |Error|assert" $OUT/rows.txt | cut -c1-600
