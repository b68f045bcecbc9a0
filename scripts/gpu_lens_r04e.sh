#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lens_r04; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_lens_rows.py -q -m gpu -s > $OUT/rows.txt 2>&1; echo "rows rc $?" >> $OUT/rows.txt
grep -E "rows/workgroup|passed|failed|rc |Error|assert" $OUT/rows.txt | cut -c1-60,200-400
timeout -k 10 900 python3 -m pytest tests/test_gpu_lens_fermat.py tests/test_gpu_full_size_properties.py tests/test_gpu_irregular_apertures.py tests/test_gpu_device_api.py tests/test_gpu_multi_device.py tests/test_gpu_dist_two_ranks.py tests/test_gpu_shoot_fuzz.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "older rc $?"
tail -5 $OUT/pytest.txt
timeout -k 10 600 python3 scripts/fuzz_lens.py --trials 100 > $OUT/fuzz.txt 2>&1; echo "fuzz rc $?"; tail -3 $OUT/fuzz.txt
