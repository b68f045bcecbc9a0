#!/bin/bash
# round-2 probe b: second instruction-cost table, the new default bench line at the driver's K, two-rank rehearsal of --gpus 2
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r02b; rm -rf "$OUT"; mkdir -p "$OUT"
$ROOT/scripts/ubench_issue2 > $OUT/ubench_issue2.txt 2>&1
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err
tail -c 3000 $OUT/bench20.json
RTUS_BENCH_ONE_GPU=1 RTUS_BENCH_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
   --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/reh2.json 2> $OUT/reh2.err
cat $OUT/reh2.json
