"""Turns the rocprofv3 databases scripts/profile_round.sh leaves under gpurun_out/prof_<tag>/ into the small
text files committed under profiles/ (run in the build container after the gpurun call):

    python scripts/profile_summarise.py r01
"""
import csv, json, os, sqlite3, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")


def db(name):
    d = os.path.join(src, name)
    f = [x for x in os.listdir(d) if x.endswith(".db")][0]
    return sqlite3.connect(os.path.join(d, f))


def kernel_stats(name, out):
    c = db(name)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(os.path.join(dst, out), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)])
    return {r[0]: r for r in rows}


def counters(name, out, like="%rtus_%"):
    c = db(name)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
                     "min(grid_size), min(workgroup_size), min(vgpr_count), min(sgpr_count) "
                     "from counters_collection where kernel_name like ? group by kernel_name, counter_name "
                     "order by kernel_name, counter_name", (like,)).fetchall()
    with open(os.path.join(dst, out), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch", "Min", "Max", "MeanDurationNs(profiled)",
                    "GridSize", "WorkgroupSize", "VGPRs", "SGPRs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 2), r[4], r[5], round(r[6], 1), r[7], r[8], r[9], r[10]])
    return {(r[0], r[1]): r[3] for r in rows}


ks = kernel_stats("kt", f"{tag}_bench_cfg2_kernel_stats.csv")
wr = counters("wr", f"{tag}_pmc_WRITE_SIZE_cfg2.csv")
rd = counters("rd", f"{tag}_pmc_FETCH_SIZE_cfg2.csv")
sq = counters("sq", f"{tag}_pmc_sq_cfg2.csv")
wr3 = counters("wr3", f"{tag}_pmc_WRITE_SIZE_cfg3.csv") if os.path.isdir(os.path.join(src, "wr3")) else {}
rd3 = counters("rd3", f"{tag}_pmc_FETCH_SIZE_cfg3.csv") if os.path.isdir(os.path.join(src, "rd3")) else {}
s0 = counters("sq_shoot0", f"{tag}_pmc_shoot_refscale_sq_compat.csv")
s1 = counters("sq_shoot1", f"{tag}_pmc_shoot_refscale_sq_fast.csv")

head = [k for k in ks if "rtus_tt_layers_kernel<2, false>" in k][0]
w_kib = wr[(head, "WRITE_SIZE")]
r_kib = rd[(head, "FETCH_SIZE")]
bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
traffic = {
    "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python3 bench.py --steps 20 --warmup 5 "
            "--graph off` (MI355X_MICROARCH.md HBM section): counters are in KiB per dispatch; on gfx950 FETCH_SIZE reports half "
            "of the bytes of a coalesced read stream, so it is doubled; WRITE_SIZE is exact for coalesced stores (here 8 B/lane: "
            "2,097,152 x 8 B = 16,384 KiB reproduced exactly, which calibrates the store side on this access pattern).",
    "kernel": "rtus_tt_layers_kernel<2, false>",
    "FETCH_SIZE_KiB_raw": round(r_kib, 2),
    "WRITE_SIZE_KiB": round(w_kib, 2),
    "cfg2_planar": int(round((2 * r_kib + w_kib) * 1024)),
    "algorithmic_bytes_per_launch": alg,
}
head3 = [k[0] for k in wr3 if "rtus_tt_layers_kernel<3, false>" in k[0]]
if head3 and (head3[0], "FETCH_SIZE") in rd3:
    traffic["cfg3_planar"] = int(round((2 * rd3[(head3[0], "FETCH_SIZE")] + wr3[(head3[0], "WRITE_SIZE")]) * 1024))
    traffic["cfg3_algorithmic_bytes_per_launch"] = 256 * 262144 * 8 + (2 * 256 + 2 * 262144) * 8
    traffic["cfg3_FETCH_SIZE_KiB_raw"] = round(rd3[(head3[0], "FETCH_SIZE")], 2)
    traffic["cfg3_WRITE_SIZE_KiB"] = round(wr3[(head3[0], "WRITE_SIZE")], 2)
json.dump(traffic, open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(dst, f"{tag}_bench_default.json"), "w"), indent=1)

print("headline kernel:", head, "calls", ks[head][1], "avg ns", round(ks[head][3], 1))
print("bench.py HIP-event mean (un-profiled) us:", bench["roofline"]["avg_launch_ms"] * 1e3, " value", bench["value"])
print("traffic:", traffic["cfg2_planar"], "vs algorithmic", alg, "| cfg3:", traffic.get("cfg3_planar"), "vs", traffic.get("cfg3_algorithmic_bytes_per_launch"))
for label, d in (("planar", sq), ("shoot compat", s0), ("shoot fast", s1)):
    for kn in sorted({k[0] for k in d}):
        g = lambda n: d.get((kn, n), float("nan"))
        wv = g("SQ_WAVES")
        print(f"{label}: {kn[:60]:60s} waves {wv:9.0f}  VALU/wave {g('SQ_INSTS_VALU')/wv:8.1f}  SALU/wave {g('SQ_INSTS_SALU')/wv:8.1f}  "
              f"VALU-active cyc/wave {4*g('SQ_ACTIVE_INST_VALU')/wv:8.0f}  wave cyc {4*g('SQ_WAVE_CYCLES')/wv:8.0f}  busy cyc {4*g('SQ_BUSY_CYCLES'):10.0f}")
