#!/bin/bash
# Scratch (GPU box): the bench line against the length of bench.py's untimed pre-run (PREWARM_MS), on copies of bench.py
for ms in 0.0 3.0 30.0 300.0 1000.0; do sed "s/^PREWARM_MS = [0-9.]*/PREWARM_MS = $ms/" bench.py > bench_pw_$ms.py; done
for r in 1 2 3; do
for ms in 0.0 3.0 30.0 300.0 1000.0; do
python3 bench_pw_$ms.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r prewarm_ms $ms  ms_per_step %.5f  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
rm -f bench_pw_*.py
