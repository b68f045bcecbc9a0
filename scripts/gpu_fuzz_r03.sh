#!/bin/bash
# long fuzz runs of the round-3 kernels (forward trace, planar tables incl. the tau-p tier via the sorted entry, lens tables)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fuzz_r03; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 1000 python3 scripts/fuzz_shoot.py 300 12345 --keep-going > $OUT/shoot_300.txt 2>&1; tail -3 $OUT/shoot_300.txt
timeout -k 10 300 python3 scripts/fuzz_layers.py 300 7 > $OUT/layers_300.txt 2>&1; tail -2 $OUT/layers_300.txt
timeout -k 10 300 python3 scripts/fuzz_layers.py 300 8 --taup > $OUT/layers_taup_300.txt 2>&1; tail -2 $OUT/layers_taup_300.txt
timeout -k 10 300 python3 scripts/fuzz_lens.py 120 7 > $OUT/lens_120.txt 2>&1; tail -2 $OUT/lens_120.txt
