"""One-line summary of a bench.py JSON line: python scripts/print_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "max|dt|", d.get("max_abs_dt_s"),
      "valu", {k: (d.get("roofline_valu") or {}).get(k) for k in ("insts_valu_per_solve", "frac_of_issue_bound")})
