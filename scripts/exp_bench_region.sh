#!/bin/bash
# Scratch (GPU box): wall-clock stamps inside bench.py's timed bracket (a patched copy of bench.py)
python3 - <<'PY'
s = open("bench.py").read()
old = """        t0 = time.perf_counter()
        ev0.record()                      # same (current) stream the library launches on
        if self.graph is not None:
            self.graph.replay()
        else:
            for s in range(self.steps):
                self.step(s)
        ev1.record()
        if tail is not None:
            tail()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0"""
new = """        t0 = time.perf_counter()
        ev0.record()                      # same (current) stream the library launches on
        ta = time.perf_counter()
        if self.graph is not None:
            self.graph.replay()
        else:
            for s in range(self.steps):
                self.step(s)
        tb = time.perf_counter()
        ev1.record()
        tc = time.perf_counter()
        if tail is not None:
            tail()
        td = time.perf_counter()
        torch.cuda.synchronize()
        te = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        print('[stamps us] ev0.record %.1f  replay() %.1f  ev1.record %.1f  tail %.1f  synchronize %.1f  total %.1f  events %.1f' % ((ta-t0)*1e6, (tb-ta)*1e6, (tc-tb)*1e6, (td-tc)*1e6, (te-td)*1e6, dt*1e6, ev0.elapsed_time(ev1)*1e3), file=sys.stderr)"""
assert old in s
open("bench_stamps.py", "w").write(s.replace(old, new))
PY
for r in 1 2 3; do python3 bench_stamps.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 >/dev/null | grep stamps; done
rm -f bench_stamps.py
