#!/bin/bash
# matcher change gate: every test that goes through the matcher + the reference-path bench lines (trace + matcher per pass)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/match_check; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_shoot_parity.py tests/test_gpu_device_api.py tests/test_gpu_dropin_script.py tests/test_gpu_edge_sizes.py tests/test_drivers.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
for wl in ref_sweep ref_scale; do
  timeout -k 10 200 python3 bench.py --workload $wl --steps 50 --warmup 5 --no-extra --no-cpu-baseline > $OUT/$wl.json 2> $OUT/$wl.err || { tail -5 $OUT/$wl.err; exit 1; }
  python3 -c "import json; d = json.loads(open('$OUT/$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done
