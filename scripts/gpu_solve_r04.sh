#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/solve_r04; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_solve.py tests/test_gpu_solve_modes.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.txt
timeout -k 10 300 python3 scripts/exp_solve_dev.py default three_launches one_lane 2>&1 | grep -v amdgpu.ids | tee $OUT/timing.txt
