#!/bin/bash
# solve timings of the working tree and of the experiment builds under variants/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/solve; mkdir -p "$OUT"
cd $ROOT
echo "== working tree"; timeout -k 10 200 python3 scripts/exp_solve_dev.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do echo "== $v"; RTUS_LIB=$ROOT/variants/librtus_$v.so timeout -k 10 200 python3 scripts/exp_solve_dev.py 2>&1 | grep -v amdgpu.ids; done
