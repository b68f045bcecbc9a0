#!/bin/bash
# round 4 planar gate: parity tests, then the headline bench line for the previous build and the working tree, alternately, both tiers
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/planar_r04; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fermat_layers.py tests/test_gpu_full_size_properties.py tests/test_gpu_irregular_apertures.py tests/test_gpu_edge_sizes.py tests/test_gpu_planar_tiers.py tests/test_gpu_multi_device.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.txt
for i in 1 2 3; do
  for tier in taup accurate; do
    for lib in prev tree; do
      if [ $lib = prev ]; then export RTUS_LIB=$ROOT/variants/librtus_prev.so; else unset RTUS_LIB; fi
      timeout -k 10 200 python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --tier $tier 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tier', d['ms_per_step'], d['roofline']['frac'], d.get('max_abs_dt_s'))"
    done
  done
done | tee $OUT/ab.txt
