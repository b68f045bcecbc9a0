#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04a; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_irregular_apertures.py tests/test_gpu_multi_device.py tests/test_gpu_lens_fermat.py tests/test_gpu_planar_tiers.py tests/test_gpu_device_api.py tests/test_gpu_lens_rows.py -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -25 $OUT/pytest.txt
bash scripts/selftest_predictor.sh; echo "selftest rc $?"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
  RTUS_LIB=$ROOT/variants/librtus_prev.so timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
  timeout -k 10 120 python3 scripts/ab_lens_f32.py 2>&1 | grep rows
done
