#!/bin/bash
# long fuzz runs of the table kernels (curved lens fp64 / fp32, planar layers) against their oracles
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fuzz; mkdir -p "$OUT"
cd $ROOT
rc=0
timeout -k 10 500 python3 scripts/fuzz_lens.py ${1:-1500} 2026 > $OUT/lens_long.txt 2>&1 || rc=1
tail -3 $OUT/lens_long.txt
timeout -k 10 500 python3 scripts/fuzz_layers.py ${2:-1500} 2026 > $OUT/layers_long.txt 2>&1 || rc=1
tail -3 $OUT/layers_long.txt
exit $rc
