#!/bin/bash
# round 4 lens gate: whole-row oracle tests of the multi-row / T-only paths (focus windows), the older lens tests, then timing A/B
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lens_r04; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_lens_rows.py -x -q -m gpu -s > $OUT/rows.txt 2>&1; echo "rows rc $?" >> $OUT/rows.txt
grep -E "rows/workgroup|passed|failed|rc |Error|assert" $OUT/rows.txt | cut -c1-400
timeout -k 10 600 python3 -m pytest tests/test_gpu_lens_fermat.py tests/test_gpu_full_size_properties.py tests/test_gpu_irregular_apertures.py tests/test_gpu_device_api.py tests/test_gpu_multi_device.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "older rc $?"
tail -3 $OUT/pytest.txt
for i in 1 2; do
  RTUS_LIB=$ROOT/variants/librtus_prev.so timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
  timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
done | tee $OUT/ab.txt
