#!/bin/bash
# long fuzz runs of the forward trace under the conditioning-aware criterion (progress lines every 25 trials)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fuzz; mkdir -p "$OUT"
cd $ROOT
rc=0
for spec in "400 12345" "150 2"; do
  set -- $spec
  timeout -k 10 880 python3 scripts/fuzz_shoot.py $1 $2 > $OUT/long_$2.txt 2>&1 || rc=1
  tail -2 $OUT/long_$2.txt
done
exit $rc
