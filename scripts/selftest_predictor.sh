#!/bin/bash
# The continuation tests must FAIL on a build whose predictor weights are perturbed (-DRTUS_EXP_BAD_PREDICTOR: the planar kernel's
# cubic weights by +2 % / -3 %, the lens kernel's quadratic weights by +5 % / -10 %) — VERDICT r03 item 2: a test that cannot fail
# tests nothing.  Run on a GPU box after scripts/build_variant.sh badpred -DRTUS_EXP_BAD_PREDICTOR (in the build container).
# Prints one line per test; exit 0 only if every one of them failed on the bad build and passed on the product build.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/selftest_predictor; mkdir -p "$OUT"
cd $ROOT
TESTS="tests/test_gpu_irregular_apertures.py::test_planar_irregular_apertures tests/test_gpu_irregular_apertures.py::test_lens_irregular_apertures tests/test_gpu_irregular_apertures.py::test_lens_continuation_is_doing_its_job tests/test_gpu_irregular_apertures.py::test_lens_many_elements_grid_consistency"
rc=0
for t in $TESTS; do
  n=$(echo $t | sed 's/.*:://')
  timeout -k 10 600 python3 -m pytest $t -x -q -m gpu > $OUT/good_$n.txt 2>&1; g=$?
  RTUS_LIB=$ROOT/variants/librtus_badpred.so timeout -k 10 600 python3 -m pytest $t -x -q -m gpu > $OUT/bad_$n.txt 2>&1; b=$?
  echo "$n: product build exit $g (want 0), perturbed predictor exit $b (want 1)"
  grep -h "^E  " $OUT/bad_$n.txt | head -2
  [ $g -eq 0 ] && [ $b -eq 1 ] || rc=1
done | tee $OUT/summary.txt
grep -q "exit 0 (want 0), perturbed predictor exit 1" $OUT/summary.txt && ! grep -vq "exit 0 (want 0), perturbed predictor exit 1\|^E  " $OUT/summary.txt
