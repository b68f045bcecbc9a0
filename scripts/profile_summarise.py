"""Turns the rocprofv3 CSV output scripts/profile_round.sh leaves under gpurun_out/prof_<tag>/ into the small files
committed under profiles/ (run in the build container after the gpurun call):

    python scripts/profile_summarise.py r03
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
# the headline kernel of the round's bench line (r03: the tau-p accuracy tier of the planar solver)
HEAD = {"r02": "rtus_tt_layers_kernel<3, false>"}.get(tag, "rtus_tt_layers_kernel<3, false, true, false>")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")


def rows_of(sub, suffix):
    out = []
    for f in glob.glob(os.path.join(src, sub, "**", f"*{suffix}"), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def kernel_rows(sub, label, like="rtus_"):
    """per-kernel duration statistics of one traced run (one workload size per run)"""
    acc = collections.defaultdict(list)
    for r in rows_of(sub, "kernel_trace.csv"):
        if like in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"],
                 r["LDS_Block_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    for (name, gx, gy, gz, vg, sg, sc, lds), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        d = sorted(d)
        out.append({"run": label, "kernel": name, "grid": f"{gx}x{gy}x{gz}", "calls": len(d), "avg_ns": round(sum(d) / len(d), 1),
                    "median_ns": d[len(d) // 2], "min_ns": d[0], "max_ns": d[-1], "vgprs": vg, "sgprs": sg, "scratch": sc, "lds": lds})
    return out


def counters(subs, like):
    """mean counter values per dispatch of the kernels matching `like`, merged over several passes
    (+ "_duration_ns:<counter>": mean dispatch duration in the pass that collected <counter>)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for sub in subs:
        for r in rows_of(sub, "counter_collection.csv"):
            if like in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[r["Kernel_Name"]]["_duration_ns:" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                meta[r["Kernel_Name"]] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, meta


def write_counters(path, table, meta):
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Counter", "MeanPerDispatch", "GridSize", "WorkgroupSize", "VGPRs", "SGPRs", "Scratch", "LDS"])
        for k in sorted(table):
            for c in sorted(table[k]):
                if not c.startswith("_duration_ns:"):
                    w.writerow([k, c, round(table[k][c], 2)] + list(meta[k]))


def write_rows(path, rows):
    if not rows:
        return
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


# ---- bench lines ---------------------------------------------------------------------------------------------
for name in ("bench_k20", "bench_default"):
    line = open(os.path.join(src, name + ".json")).read().strip().splitlines()[-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)
bench = json.load(open(os.path.join(dst, f"{tag}_bench_k20.json")))

# ---- kernel trace of the driver's command ------------------------------------------------------------------------
kt = kernel_rows("kt", "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline (rocprofv3 --kernel-trace --stats)")
head = [r for r in kt if HEAD in r["kernel"]][0]
# the K timed launches alone: the headline kernel's dispatches in time order are W warm-up launches, the ~30 ms of untimed
# graph replays (clock ramp), the K timed ones and one more inside `extra`; the driver's command has K = 20
hd = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows_of("kt", "kernel_trace.csv")
            if HEAD in r["Kernel_Name"])
K = int(bench["steps"])
timed = sorted(d for _, d in hd[-(K + 1):-1])
if len(timed) == K:
    kt.insert(0, dict(head, run=head["run"] + f" - the {K} TIMED launches only (the last {K} dispatches of the graph replays, before `extra`)",
                      calls=K, avg_ns=round(sum(timed) / K, 1), median_ns=timed[K // 2], min_ns=timed[0], max_ns=timed[-1]))
write_rows(os.path.join(dst, f"{tag}_bench_kernel_trace.csv"), kt)

# ---- HBM traffic of the headline launch --------------------------------------------------------------------------
wr, m1 = counters(["wr"], HEAD)
rd, m2 = counters(["rd"], HEAD)
write_counters(os.path.join(dst, f"{tag}_pmc_WRITE_SIZE_cfg3.csv"), wr, m1)
write_counters(os.path.join(dst, f"{tag}_pmc_FETCH_SIZE_cfg3.csv"), rd, m2)
hk = list(wr)[0]
w_kib, r_kib = wr[hk]["WRITE_SIZE"], rd[hk]["FETCH_SIZE"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
traffic = {
    "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python3 bench.py --steps 5 --warmup 2 --graph off "
            "--no-extra --no-cpu-baseline` (MI355X_MICROARCH.md HBM section): counters are KiB per dispatch; on gfx950 FETCH_SIZE reports "
            "half of the bytes of a coalesced read stream, so it is doubled; WRITE_SIZE is exact for coalesced stores (67,108,864 x 8 B = "
            "524,288 KiB reproduced exactly, which calibrates the store side on this access pattern).",
    "kernel": hk, "FETCH_SIZE_KiB_raw": round(r_kib, 2), "WRITE_SIZE_KiB": round(w_kib, 2),
    "cfg3_planar": int(round((2 * r_kib + w_kib) * 1024)), "algorithmic_bytes_per_launch": alg,
}
cw, _ = counters(["wr_cons"], "rtus_")
cr, _ = counters(["rd_cons"], "rtus_")
for k in cw:
    short = "tfm" if "tfm" in k else ("focal_delays" if "focal" in k else None)
    if short and k in cr:
        traffic[short + "_bytes_per_launch"] = int(round((2 * cr[k]["FETCH_SIZE"] + cw[k]["WRITE_SIZE"]) * 1024))
        traffic[short + "_FETCH_SIZE_KiB_raw"] = round(cr[k]["FETCH_SIZE"], 1)
        traffic[short + "_WRITE_SIZE_KiB"] = round(cw[k]["WRITE_SIZE"], 1)
json.dump(traffic, open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1)

# ---- SQ counters: headline kernel and the other table kernels ------------------------------------------------------
sq, meta = counters(["sq", "sq2", "sq3"], HEAD)
write_counters(os.path.join(dst, f"{tag}_pmc_sq_cfg3.csv"), sq, meta)
valu = {}
c = sq[hk]
waves = c["SQ_WAVES"]
solves = 256 * 262144
wave_solves = solves / 64.0
f64 = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]
trans = c["SQ_INSTS_VALU_TRANS_F32"]
cvt = c["SQ_INSTS_VALU_CVT"]
f32 = c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]
other = c["SQ_INSTS_VALU"] - f64 - trans - cvt - f32
# issue cycles per wave-instruction measured by scripts/ubench_issue*.hip (8 waves per SIMD): fp64 4.2, v_rsq/v_rcp_f32 8.2,
# cvt 4.2, fp32 fma/mul/add with VGPR operands 2.3, everything else (moves, bit ops, compares, max) 2.3 .. 4.2 -> 3.3
issue = f64 * 4.2 + trans * 8.2 + cvt * 4.2 + f32 * 2.3 + other * 3.3
kern_ns = head["avg_ns"]
# effective clock of the profiled launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
clock = c["GRBM_GUI_ACTIVE"] / 8.0 / c["_duration_ns:GRBM_GUI_ACTIVE"]
sq_ns = c["_duration_ns:SQ_WAVE_CYCLES"]
valu["cfg3_planar"] = {
    "kernel": hk,
    "insts_valu_per_solve": round(c["SQ_INSTS_VALU"] / wave_solves, 2),
    "insts_valu_per_solve_by_class": {"fp64_fma_mul_add": round(f64 / wave_solves, 2), "fp32_fma_mul_add": round(f32 / wave_solves, 2),
                                      "trans_f32": round(trans / wave_solves, 2), "cvt": round(cvt / wave_solves, 2),
                                      "other": round(other / wave_solves, 2)},
    "insts_salu_per_solve": round(c["SQ_INSTS_SALU"] / wave_solves, 2),
    "insts_lds_per_solve": round(c.get("SQ_INSTS_LDS", 0) / wave_solves, 2),
    "issue_cycles_per_wave_solve": round(issue / wave_solves, 1),
    "issue_cycles_how": "instruction classes (SQ_INSTS_VALU_*) x issue cycles per wave-instruction measured by scripts/ubench_issue*.hip "
                        "(profiles/r02_ubench_issue.txt)",
    "valu_active_frac_of_wave_life": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4),
    "waves": int(waves), "waves_per_simd_launched": round(waves / 1024.0, 1),
    "valu_active_x_resident_waves": round(8 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3),
    "wave_life_us_mean": round(4 * c["SQ_WAVE_CYCLES"] / waves / (clock * 1e3), 2),
    "wave_slots_occupied_frac_of_kernel": round((4 * c["SQ_WAVE_CYCLES"] / (clock * 1e9)) / (8 * 1024 * sq_ns * 1e-9), 3),
    "profiled_launch_us": round(sq_ns / 1e3, 1),
    "clock_ghz": round(clock, 3),
    "clock_how": "GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration of the counter pass (reads a few % high on dispatches this short)",
}
# round 4: the same kernel as a PERSISTENT grid (-DRTUS_EXP_PERSIST, variants/librtus_persist.so): are half-empty wave slots the headline's bound?
if os.path.isdir(os.path.join(src, "sq_persist")):
    tp, mp = counters(["sq_persist"], HEAD)
    cp = next(iter(tp.values())) if tp else {}
    if cp.get("SQ_WAVE_CYCLES"):
        clk = clock
        ns = cp["_duration_ns:SQ_WAVE_CYCLES"]
        valu["cfg3_planar_persistent_grid_experiment"] = {
            "build": "-DRTUS_EXP_PERSIST: 2,048 workgroups (8 per CU, every wave slot taken from start to end), each takes work items w, w + 2048, ... (the product launches 4,096 workgroups, two rounds of 8 per CU)",
            "waves": int(cp["SQ_WAVES"]), "insts_valu": int(cp["SQ_INSTS_VALU"]),
            "wave_life_us_mean": round(4 * cp["SQ_WAVE_CYCLES"] / cp["SQ_WAVES"] / (clk * 1e3), 2),
            "wave_slots_occupied_frac_of_kernel": round((4 * cp["SQ_WAVE_CYCLES"] / (clk * 1e9)) / (8 * 1024 * ns * 1e-9), 3),
            "profiled_launch_us": round(ns / 1e3, 1),
            "product_profiled_launch_us": valu["cfg3_planar"]["profiled_launch_us"],
            "product_wave_slots_occupied_frac_of_kernel": valu["cfg3_planar"]["wave_slots_occupied_frac_of_kernel"],
            "reading": "full wave slots do not shorten the launch (launch shapes of 3 .. 8 workgroups per CU were measured too: rtus_fermat.hip launch_layers, DESIGN.md section 4): the kernel is bound by VALU issue, not by residency",
        }
        write_counters(os.path.join(dst, f"{tag}_pmc_sq_cfg3_persist.csv"), tp, mp)
json.dump(valu, open(os.path.join(dst, f"valu_{tag}.json"), "w"), indent=1)
for wl, like in (("cfg4_lens_f32", "rtus_tt_lens_kernel"), ("cfg2_planar", "rtus_tt_layers_kernel"), ("cfg5_fmc", "rtus_tt_layers_kernel")):
    t, m = counters([f"sq_{wl}"], like)
    write_counters(os.path.join(dst, f"{tag}_pmc_sq_{wl}.csv"), t, m)

# ---- forward trace: one size per row -------------------------------------------------------------------------------
shoot = (kernel_rows("kt_shoot0", "reference geometry 1024 tx x 8192 rays, reference-compatible arithmetic (scripts/run_shoot_once.py 0)") +
         kernel_rows("kt_shoot1", "reference geometry 1024 tx x 8192 rays, vector-form arithmetic (scripts/run_shoot_once.py 1)") +
         kernel_rows("kt_sweep", "the reference's own sweep, 210 geometries x 905 rays (bench.py --workload ref_sweep --graph off: the fused entry)") +
         kernel_rows("kt_sweep_two", "the same sweep as two calls, rtus_shoot_dev + rtus_match_dev (scripts/run_sweep_once.py two)") +
         kernel_rows("kt_sweep_fused", "the same sweep through rtus_sweep_dev (scripts/run_sweep_once.py fused)") +
         kernel_rows("kt_sweep_kept", "the same sweep through rtus_sweep_dev with the polyline kept, RTUS_POLYLINE_READY (scripts/run_sweep_once.py kept)"))
write_rows(os.path.join(dst, f"{tag}_shoot_kernel_rows.csv"), shoot)
for mode, name in ((0, "compat"), (1, "fast")):
    t, m = counters([f"sq_shoot{mode}", f"sq_shoot{mode}b"], "rtus_shoot_kernel")
    write_counters(os.path.join(dst, f"{tag}_pmc_shoot_refscale_sq_{name}.csv"), t, m)
    for k, cs in t.items():
        print(f"shoot {name}: VALU/wave {cs['SQ_INSTS_VALU'] / cs['SQ_WAVES']:.0f}  SALU/wave {cs['SQ_INSTS_SALU'] / cs['SQ_WAVES']:.0f}  "
              f"SMEM/wave {cs.get('SQ_INSTS_SMEM', 0) / cs['SQ_WAVES']:.0f}")
write_rows(os.path.join(dst, f"{tag}_consumers_kernel_rows.csv"), kernel_rows("kt_cons", "scripts/run_consumers_once.py"))

# ---- instruction issue costs (round 2's tables; later rounds re-use them) ------------------------------------------------
if os.path.exists(os.path.join(src, "ubench_issue.txt")):
    with open(os.path.join(dst, f"{tag}_ubench_issue.txt"), "w") as fh:
        for n in ("ubench_issue", "ubench_issue2", "ubench_issue3", "ubench_issue4"):
            fh.write(f"==== scripts/{n}.hip ====\n" + open(os.path.join(src, n + ".txt")).read() + "\n")

# ---- issue-bound objects of the other kernels bench.py prints a roofline_valu for -----------------------------------------
def clock_of(c):
    """effective shader clock of the counter pass; dispatches shorter than ~0.1 ms read far too high (the counter keeps running
    between a short dispatch's end and its read-out): the nominal 2.1 GHz there"""
    if "GRBM_GUI_ACTIVE" not in c or c["_duration_ns:GRBM_GUI_ACTIVE"] < 1.0e5:
        return 2.1
    return c["GRBM_GUI_ACTIVE"] / 8.0 / c["_duration_ns:GRBM_GUI_ACTIVE"]


# the multi-GPU headline's kernel: configs[3] shard of 8 (128 rows x 1024^2 targets, fp32)
tl, ml = counters(["sq_cfg4_lens_f32", "sq_cfg4_lens_f32_b", "sq_cfg4_lens_f32_c"], "rtus_tt_lens_kernel")
if tl:
    lk = [k for k in tl if "float" in k][0]
    c = tl[lk]
    ws = 1024 * 1048576 / 64.0                      # bench.py --workload cfg4_lens_f32 on ONE GPU solves the whole 1024-row table
    f32 = c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c.get("SQ_INSTS_VALU_ADD_F32", 0)
    tr = c.get("SQ_INSTS_VALU_TRANS_F32", 0)
    cv = c.get("SQ_INSTS_VALU_CVT", 0)
    f64 = c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)
    oth = c["SQ_INSTS_VALU"] - f32 - tr - cv - f64
    # fp32 FMA-class with VGPR operands 2.3 cycles, transcendental 8.2, cvt / fp64 4.2, the rest (compares, selects, bit ops) ~3.3;
    # the scalar ALU issues beside the VALU for free up to one scalar per vector instruction (profiles/r02_ubench_issue.txt)
    issue = f32 * 2.3 + tr * 8.2 + (cv + f64) * 4.2 + oth * 3.3
    valu["cfg4_lens_f32"] = {
        "kernel": lk, "insts_valu_per_solve": round(c["SQ_INSTS_VALU"] / ws, 2), "insts_salu_per_solve": round(c["SQ_INSTS_SALU"] / ws, 2),
        "insts_valu_per_solve_by_class": {"fp32_fma_mul_add": round(f32 / ws, 2), "trans_f32": round(tr / ws, 2), "cvt": round(cv / ws, 2),
                                          "fp64": round(f64 / ws, 2), "other": round(oth / ws, 2)},
        "issue_cycles_per_wave_solve": round(issue / ws, 1), "clock_ghz": round(clock_of(c), 3),
        "valu_active_frac_of_wave_life": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4),
        "workload_of_the_counters": "the whole BASELINE configs[3] table on one GPU: 1024 tx rows x 1024 x 1024 targets",
        "issue_cycles_how": "instruction classes x issue cycles per wave-instruction of profiles/r02_ubench_issue.txt"}

# forward trace, reference geometry 1024 tx x 8192 rays
for mode, key in ((0, "ref_scale"), (1, "ref_scale_fastmath")):
    t, m = counters([f"sq_shoot{mode}", f"sq_shoot{mode}b"], "rtus_shoot_kernel")
    for k, c in t.items():
        w = c["SQ_WAVES"]
        fl = 2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)
        valu[key] = {"kernel": k, "waves": int(w), "insts_valu_per_wave": round(c["SQ_INSTS_VALU"] / w, 1),
                     "insts_salu_per_wave": round(c["SQ_INSTS_SALU"] / w, 1), "insts_smem_per_wave": round(c.get("SQ_INSTS_SMEM", 0) / w, 1),
                     "fp64_arith_insts_per_wave": round((c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) +
                                                         c.get("SQ_INSTS_VALU_MUL_F64", 0)) / w, 1),
                     "issue_cycles_per_pass": round(c["SQ_INSTS_VALU"] * 4.2, 0), "fp64_flop_per_pass": round(fl * 64, 0),
                     "clock_ghz": round(clock_of(c), 3),
                     "valu_active_frac_of_wave_life": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4),
                     "issue_cycles_how": "4.2 issue cycles per VALU wave-instruction (fp64 and everything beside it in this kernel: "
                                         "compares, selects, bit operations; profiles/r02_ubench_issue.txt)"}

# root-finding solve: both kernels of a pass
solve_rows = []
for size in ("sweep", "scale"):
    for mode, suffix in (("compat", ""), ("fast", "_fastmath")):
        t, m = counters([f"sq_solve_{size}_{mode}", f"sq_solve_{size}_{mode}b"], "rtus_s")
        if not t:
            continue
        write_counters(os.path.join(dst, f"{tag}_pmc_solve_{size}_{mode}.csv"), t, m)
        solve_rows += kernel_rows(f"kt_solve_{size}_{mode}", f"scripts/run_solve_once.py {size} {mode}")
        per = {}
        tot_valu = fl = 0.0
        clk = 2.1
        for k, c in t.items():
            short = "grid_trace" if "shoot" in k else ("grid_trace_and_refine_one_launch" if "row_kernel" in k else "refine")
            w = c["SQ_WAVES"]
            per[short] = {"kernel": k, "waves": int(w), "insts_valu_per_wave": round(c["SQ_INSTS_VALU"] / w, 1),
                          "insts_salu_per_wave": round(c["SQ_INSTS_SALU"] / w, 1), "insts_smem_per_wave": round(c.get("SQ_INSTS_SMEM", 0) / w, 1),
                          "insts_valu": int(c["SQ_INSTS_VALU"]), "valu_active_frac_of_wave_life": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4)}
            tot_valu += c["SQ_INSTS_VALU"]
            fl += (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)) * 64
            clk = clock_of(c) if "shoot" in k else clk                  # the longer of the two dispatches
        valu[f"solve_{size}{suffix}"] = {"kernels": per, "issue_cycles_per_pass": round(tot_valu * 4.2, 0), "fp64_flop_per_pass": round(fl, 0),
                                         "clock_ghz": round(clk, 3),
                                         "issue_cycles_how": "VALU wave-instructions of both kernels x 4.2 issue cycles, spread over 1024 SIMDs"}
write_rows(os.path.join(dst, f"{tag}_solve_kernel_rows.csv"), solve_rows)
json.dump(valu, open(os.path.join(dst, f"valu_{tag}.json"), "w"), indent=1)

print("headline kernel:", hk, "| kernel-trace avg ns", head["avg_ns"], "median", head["median_ns"], "calls", head["calls"])
print("bench.py (un-profiled, K=20): value", bench["value"], "ms/step", bench["ms_per_step"], "avg_launch_ms", bench["roofline"]["avg_launch_ms"],
      "frac", bench["roofline"]["frac"])
print("traffic:", traffic["cfg3_planar"], "vs algorithmic", alg, "->", round(traffic["cfg3_planar"] / alg, 4))
print("valu:", json.dumps(valu["cfg3_planar"], indent=1))
print({k: v for k, v in traffic.items() if k.startswith(("tfm", "focal"))})
