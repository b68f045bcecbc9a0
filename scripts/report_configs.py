"""Per-config report (BASELINE.md "Reported per config"): for each BASELINE config + the reference-parity scale-up,
throughput / launch time / HBM fraction from bench.py, and max |dt| of the GPU result against the CPU restatement
(oracle/) on a bounded sample of the same inputs.  Writes profiles/<tag>_config_report.{json,md}.

    gpurun -- python scripts/report_configs.py r01
"""
import json, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
import bench
import rtus
from oracle import cport


def run_bench(wl):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-extra", "--no-cpu-baseline",
                        "--steps", "2000" if wl in ("cfg2_planar", "cfg5_fmc", "ref_sweep") else "100"],
                       capture_output=True, text=True)
    return json.loads(r.stdout.strip().splitlines()[-1])


def max_dt(wl):
    """max |dt| GPU vs oracle on a sample (elements x targets small enough for the CPU checker)."""
    if wl in ("cfg2_planar", "cfg3_planar"):
        W = bench.planar_inputs(wl, 0, 1)
        e = np.linspace(0, W["n_e"] - 1, 5).astype(int)
        f = np.linspace(0, W["n_f"] - 1, 4096).astype(int)
        a = (W["z_if"], W["c"], W["xe"][e], W["ze"][e], W["xf"][f], W["zf"][f])
        return float(np.nanmax(np.abs(rtus.travel_time_layers(*a) - cport.tt_layers(*a)))), "5 elements x 4096 focal points vs orc_tt_layers (long double)"
    if wl == "cfg5_fmc":
        W = bench.fmc_inputs(0, 1)
        e = np.linspace(0, W["n_e"] - 1, 8).astype(int)
        a = (W["z_if"], W["c"], W["xe"][e], W["ze"][e], W["xf"], W["zf"])
        return float(np.nanmax(np.abs(rtus.travel_time_layers(*a) - cport.tt_layers(*a)))), "8 tx rows x 2048 rx vs orc_tt_layers (long double)"
    if wl == "cfg4_lens_f32":
        W = bench.lens_inputs(0, 1)
        e = np.linspace(0, W["n_e"] - 1, 3).astype(int)
        f = np.linspace(0, W["n_f"] - 1, 2048).astype(int)
        ref, _ = cport.tt_lens(W["xe"][e], W["ze"][e], W["xf"][f], W["zf"][f], -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
        g32 = rtus.travel_time_lens(W["xe"][e], W["ze"][e], W["xf"][f], W["zf"][f], params=rtus.Params(), dtype=np.float32)
        g64 = rtus.travel_time_lens(W["xe"][e], W["ze"][e], W["xf"][f], W["zf"][f], params=rtus.Params())
        return float(np.nanmax(np.abs(g32 - ref))), (f"3 elements x 2048 targets, fp32 vs orc_tt_lens (long double); fp64 kernel on the same "
                                                    f"sample: {float(np.nanmax(np.abs(g64 - ref))):.1e} s")
    R = bench.ref_inputs(wl)
    g = R["geoms"][:: max(1, R["geoms"].shape[0] // 4)][:4]
    xa = R["xa"][:: max(1, R["xa"].size // 3)][:3]
    b = rtus.shoot_batch(xa, np.full(xa.size, R["za"][0]), R["zf"], R["alpha"], g, params=rtus.Params(), want=("tof",))
    worst = 0.0
    for gi in range(g.shape[0]):
        for t in range(xa.size):
            o, _ = cport.shoot(xa[t], R["za"][0], R["zf"], R["alpha"], g[gi, 0], g[gi, 1])
            t4 = cport.tof4(xa[t], R["za"][0], o)
            tof = ((t4[0] + t4[1]) + t4[2]) + t4[3]
            assert np.array_equal(np.isnan(tof), np.isnan(b["tof"][gi, t]))
            m = ~np.isnan(tof)
            worst = max(worst, float(np.max(np.abs(tof - b["tof"][gi, t])[m])) if m.any() else 0.0)
    return worst, f"{g.shape[0]} geometries x {xa.size} tx x {R['n']} rays vs orc_shoot (NaN masks identical)"


rows = []
for wl in ("cfg2_planar", "cfg3_planar", "cfg4_lens_f32", "cfg5_fmc", "ref_sweep", "ref_scale"):
    d = run_bench(wl)
    dt, how = max_dt(wl)
    rows.append({"workload": wl, "config": d["config"]["workload"], "dtype": d["dtype"], "Mrays_per_s": d["value"],
                 "ms_per_step": d["ms_per_step"], "hbm_frac": d["roofline"]["frac"], "hbm_GBps": d["roofline"]["achieved"],
                 "kernel": d["roofline"]["kernel"], "max_abs_dt_s": dt, "dt_sample": how})
    print(rows[-1], flush=True)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
json.dump(rows, open(os.path.join(out, f"{tag}_config_report.json"), "w"), indent=1)
with open(os.path.join(out, f"{tag}_config_report.md"), "w") as fh:
    fh.write("| workload | dtype | Mrays/s (1 GPU) | ms/step | HBM GB/s | % of 8 TB/s | max \\|dt\\| vs CPU restatement | sample |\n|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write(f"| {r['config']} | {r['dtype']} | {r['Mrays_per_s']:.0f} | {r['ms_per_step']:.4f} | {r['hbm_GBps']:.0f} | "
                 f"{100 * r['hbm_frac']:.1f} | {r['max_abs_dt_s']:.1e} s | {r['dt_sample']} |\n")
print(open(os.path.join(out, f"{tag}_config_report.md")).read())
