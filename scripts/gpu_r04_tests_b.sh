#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04b; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_sweep_random_rows.py tests/test_gpu_sweep_fused.py tests/test_gpu_shoot_fuzz.py tests/test_gpu_multi_device.py tests/test_gpu_dist_two_ranks.py -q -m gpu -s > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; grep -E "rows x 65|SECOND LOOKS|passed|failed|Error" $OUT/pytest.txt | cut -c1-400; grep -c "^geom " $OUT/pytest.txt
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; tail -3 $OUT/bench_default.err
python3 scripts/print_bench.py $OUT/bench_default.json 2>/dev/null | head -60
