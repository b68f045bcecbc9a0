#!/bin/bash
# change gate of the root-finding solve: its tests, the forward-trace parity tests (they share trace_ray), then timings
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/solve; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_solve.py tests/test_gpu_solve_modes.py tests/test_gpu_shoot_parity.py tests/test_gpu_edge_sizes.py tests/test_gpu_shoot_fuzz.py tests/test_drivers.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 300 python3 scripts/exp_solve_dev.py > $OUT/exp_solve_dev.log 2>&1 || { tail -30 $OUT/exp_solve_dev.log; exit 1; }
cat $OUT/exp_solve_dev.log
