#!/bin/bash
# long fuzz runs of round 4's kernels: the lens tables (least time over the whole interval), the planar kernel, the forward trace
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fuzz_r04; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 400 python3 scripts/fuzz_lens.py 1000 2026 > $OUT/lens_long.txt 2>&1; echo "lens rc $?"; tail -2 $OUT/lens_long.txt
timeout -k 10 300 python3 scripts/fuzz_layers.py 600 2026 > $OUT/layers_long.txt 2>&1; echo "layers rc $?"; tail -2 $OUT/layers_long.txt
timeout -k 10 500 python3 scripts/fuzz_shoot.py 250 20264 > $OUT/shoot_long.txt 2>&1; echo "shoot rc $?"; tail -2 $OUT/shoot_long.txt
