#!/bin/bash
# whole GPU suite + the default bench line (what the driver runs at round end)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/full; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err || { tail -20 $OUT/bench20.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "max|dt|", d.get("max_abs_dt_s"))
for k, v in d.get("extra", {}).items():
    print(" ", k, {a: b for a, b in v.items() if a not in ("note", "roofline")}, (v.get("roofline") or {}).get("G_rays_per_s_kernel_only"))
PY
