for r in 1 2 3; do
for w in 5 50 500; do
python3 bench.py --steps 20 --warmup $w --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r warmup $w  ms_per_step %.5f  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
