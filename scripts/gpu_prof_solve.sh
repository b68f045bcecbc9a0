#!/bin/bash
# kernel trace of the solve pipeline at both sizes (rocprofv3 --kernel-trace), arithmetic mode = $1 (compat | fast)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=${1:-compat}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_solve_sweep $ROOT/gpurun_out/prof_solve_scale
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_solve_sweep -o kt -- python3 $ROOT/scripts/run_solve_once.py sweep $MODE 20 > $ROOT/gpurun_out/prof_solve_sweep.log 2>&1 &&
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_solve_scale -o kt -- python3 $ROOT/scripts/run_solve_once.py scale $MODE 5 > $ROOT/gpurun_out/prof_solve_scale.log 2>&1 &&
python3 $ROOT/scripts/kt_summary.py $ROOT/gpurun_out/prof_solve_sweep $ROOT/gpurun_out/prof_solve_scale
