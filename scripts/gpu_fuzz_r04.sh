#!/bin/bash
# long fuzz runs of round 4's kernels: the lens tables (least time over the whole interval), the planar kernel in both tiers (every
# fourth trial a table with >= 8 rows per workgroup), the forward trace, the root-finding solve in its three forms
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fuzz_r04; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 320 python3 scripts/fuzz_lens.py 500 2026 > $OUT/lens_long.txt 2>&1; echo "lens rc $?"; tail -1 $OUT/lens_long.txt
timeout -k 10 200 python3 scripts/fuzz_layers.py 600 2026 > $OUT/layers_long.txt 2>&1; echo "layers rc $?"; tail -1 $OUT/layers_long.txt
timeout -k 10 200 python3 scripts/fuzz_layers.py 600 2028 --taup > $OUT/layers_taup_long.txt 2>&1; echo "layers tau-p rc $?"; tail -1 $OUT/layers_taup_long.txt
timeout -k 10 320 python3 scripts/fuzz_shoot.py 150 20264 > $OUT/shoot_long.txt 2>&1; echo "shoot rc $?"; tail -1 $OUT/shoot_long.txt
timeout -k 10 120 python3 scripts/fuzz_solve.py 420 99 > $OUT/solve_long.txt 2>&1; echo "solve rc $?"; tail -1 $OUT/solve_long.txt
