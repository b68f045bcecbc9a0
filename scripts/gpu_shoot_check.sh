#!/bin/bash
# forward-trace change gate: parity tests that touch the trace, a conditioning-aware fuzz, then an interleaved A/B against
# a previous build:  bash scripts/gpu_shoot_check.sh [old.so] [fuzz trials]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/shoot_check; mkdir -p "$OUT"
cd $ROOT
OLD=${1:-scripts/librtus_v2c.so}; N=${2:-40}
timeout -k 10 600 python3 -m pytest tests/test_gpu_shoot_parity.py tests/test_gpu_solve.py tests/test_gpu_solve_modes.py tests/test_gpu_edge_sizes.py tests/test_gpu_dropin_script.py tests/test_drivers.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 600 python3 scripts/fuzz_shoot.py $N 31337 > $OUT/fuzz.txt 2>&1 || { tail -12 $OUT/fuzz.txt; exit 1; }
tail -3 $OUT/fuzz.txt
timeout -k 10 300 python3 scripts/ab_shoot_variants.py prev > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
grep -v "amdgpu.ids" $OUT/ab.txt
