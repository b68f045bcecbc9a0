#!/bin/bash
# Scratch (GPU box): the bench line at the driver's K with the K launches captured into a hipGraph or launched eagerly (wall clock per step vs the HIP-event mean)
for r in 1 2 3; do
for g in on off; do
python3 bench.py --steps 20 --warmup 5 --graph $g --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r graph $g  wall ms_per_step %.5f  event ms %.5f  overhead per region %.1f us' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], (d['ms_per_step'] - d['roofline']['avg_launch_ms']) * 20 * 1e3))"
done; done
