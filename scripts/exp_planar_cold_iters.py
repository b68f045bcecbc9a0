"""Scratch (GPU box): Newton steps per row of a block of BASELINE configs[2] (the four cold rows against the four-history rows)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import rtus, bench
W = bench.workload_inputs("cfg3_planar", 0, 1)
sel = np.arange(0, W["n_f"], 16)                       # every 16th target
tt, it = rtus.travel_time_layers(W["z_if"], W["c"], W["xe"], W["ze"], W["xf"][sel], W["zf"][sel], return_iters=True)
print("rows per block", rtus.lib().rtus_table_rows_per_block(W["n_e"], sel.size, 8))
for r in range(12):
    v = it[r]
    print(f"row {r}: mean Newton steps {v.mean():.3f}  max {v.max()}  share with >0 steps {(v > 0).mean():.4f}")
print("rows 4..63: mean", it[4:].mean())
