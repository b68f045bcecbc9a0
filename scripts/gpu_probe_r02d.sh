#!/bin/bash
# round-2 probe d: new planar kernel — parity tests, A/B against the round-1 build, bench line
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r02d; rm -rf "$OUT"; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fermat_layers.py tests/test_gpu_full_size_properties.py tests/test_gpu_irregular_apertures.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 300 python3 scripts/ab_planar.py scripts/librtus_r01.so ray-tracing-ultrasound_amd/librtus.so > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
cat $OUT/ab.txt
