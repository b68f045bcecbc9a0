#!/bin/bash
# planar-layer Fermat kernel change gate: its parity / fuzz / full-size tests, then an interleaved A/B against a previous build
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/planar_check; mkdir -p "$OUT"
cd $ROOT
OLD=${1:-scripts/librtus_v6a.so}
timeout -k 10 600 python3 -m pytest tests/test_gpu_fermat_layers.py tests/test_gpu_full_size_properties.py tests/test_gpu_irregular_apertures.py tests/test_gpu_edge_sizes.py tests/test_gpu_tfm.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 300 python3 scripts/ab_planar.py $OLD ray-tracing-ultrasound_amd/librtus.so ${2:-7} > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
grep -v "amdgpu.ids" $OUT/ab.txt
