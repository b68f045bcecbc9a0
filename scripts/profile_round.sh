#!/bin/bash
# Re-creates the rocprofv3 evidence under profiles/ for one round on the GPU box:
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r02'
# Raw rocprofv3 output lands in gpurun_out/prof_<tag>/; scripts/profile_summarise.py turns it into profiles/<tag>_*.
# rocprofv3 rules of this pool: run from /tmp with TMPDIR=/tmp, the program itself right after `--`, counters in their
# own passes (never together with a trace domain other than --kernel-trace).
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
B=$ROOT/bench.py
PMC="--steps 5 --warmup 2 --graph off --no-extra --no-cpu-baseline"       # headline workload (cfg3_planar), eager launches
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
SQSH="SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
SQ3="SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY GRBM_GUI_ACTIVE"
echo "[1] un-profiled bench lines: the driver's K (20 / 5) and the default K"
python3 $B --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
python3 $B --no-extra --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "[2] kernel trace of the driver's command"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1
echo "[3] HBM traffic of the headline launch (separate passes)"
rocprofv3 --pmc WRITE_SIZE -d $OUT/wr -o wr --output-format csv -- python3 $B $PMC > $OUT/wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/rd -o rd --output-format csv -- python3 $B $PMC > $OUT/rd.log 2>&1
echo "[4] SQ counters of the headline kernel"
rocprofv3 --pmc $SQ -d $OUT/sq -o sq --output-format csv -- python3 $B $PMC > $OUT/sq.log 2>&1
rocprofv3 --pmc $SQ2 -d $OUT/sq2 -o sq --output-format csv -- python3 $B $PMC > $OUT/sq2.log 2>&1
rocprofv3 --pmc $SQ3 -d $OUT/sq3 -o sq --output-format csv -- python3 $B $PMC > $OUT/sq3.log 2>&1
if [ -f $ROOT/variants/librtus_persist.so ]; then
echo "[4b] the headline kernel as a persistent grid (experiment build): SQ counters + its bench line"
export RTUS_LIB=$ROOT/variants/librtus_persist.so
rocprofv3 --pmc $SQ -d $OUT/sq_persist -o sq --output-format csv -- python3 $B $PMC > $OUT/sq_persist.log 2>&1
python3 $B --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_persist.json 2> $OUT/bench_persist.err
unset RTUS_LIB
fi
echo "[5] the multi-GPU headline's kernel on one GPU (configs[3] shard of 8: 128 rows) and configs[1]"
for wl in cfg4_lens_f32 cfg2_planar cfg5_fmc; do
  rocprofv3 --pmc $SQ -d $OUT/sq_$wl -o sq --output-format csv -- python3 $B --workload $wl $PMC > $OUT/sq_$wl.log 2>&1
done
rocprofv3 --pmc $SQ2 -d $OUT/sq_cfg4_lens_f32_b -o sq --output-format csv -- python3 $B --workload cfg4_lens_f32 $PMC > $OUT/sq_cfg4_lens_f32_b.log 2>&1
rocprofv3 --pmc $SQ3 -d $OUT/sq_cfg4_lens_f32_c -o sq --output-format csv -- python3 $B --workload cfg4_lens_f32 $PMC > $OUT/sq_cfg4_lens_f32_c.log 2>&1
echo "[6] forward trace: one size per run (reference geometry 1024 tx x 8192 rays; the reference's own sweep)"
for mode in 0 1; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_shoot$mode -o kt --output-format csv -- python3 $ROOT/scripts/run_shoot_once.py $mode > $OUT/kt_shoot$mode.log 2>&1
  rocprofv3 --pmc $SQ -d $OUT/sq_shoot$mode -o sq --output-format csv -- python3 $ROOT/scripts/run_shoot_once.py $mode > $OUT/sq_shoot$mode.log 2>&1
  rocprofv3 --pmc $SQSH -d $OUT/sq_shoot${mode}b -o sq --output-format csv -- python3 $ROOT/scripts/run_shoot_once.py $mode > $OUT/sq_shoot${mode}b.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/kt_sweep -o kt --output-format csv -- python3 $B --workload ref_sweep --steps 50 --warmup 5 --graph off --no-extra --no-cpu-baseline > $OUT/kt_sweep.log 2>&1
for form in two fused kept; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_sweep_$form -o kt --output-format csv -- python3 $ROOT/scripts/run_sweep_once.py $form > $OUT/kt_sweep_$form.log 2>&1
done
echo "[7] consumers: traffic of the TFM gather kernel and of the focal-law stream"
rocprofv3 --pmc FETCH_SIZE -d $OUT/rd_cons -o rd --output-format csv -- python3 $ROOT/scripts/run_consumers_once.py > $OUT/rd_cons.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/wr_cons -o wr --output-format csv -- python3 $ROOT/scripts/run_consumers_once.py > $OUT/wr_cons.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_cons -o kt --output-format csv -- python3 $ROOT/scripts/run_consumers_once.py > $OUT/kt_cons.log 2>&1
echo "[8] root-finding solve: kernel trace + SQ counters per size and arithmetic mode (one size per run)"
SQSOLVE="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"
SQSOLVE2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE"
for size in sweep scale; do
  for mode in compat fast; do
    reps=20; [ $size = scale ] && reps=5
    rocprofv3 --kernel-trace --stats -d $OUT/kt_solve_${size}_$mode -o kt --output-format csv -- python3 $ROOT/scripts/run_solve_once.py $size $mode $reps > $OUT/kt_solve_${size}_$mode.log 2>&1
    rocprofv3 --pmc $SQSOLVE -d $OUT/sq_solve_${size}_$mode -o sq --output-format csv -- python3 $ROOT/scripts/run_solve_once.py $size $mode 4 > $OUT/sq_solve_${size}_$mode.log 2>&1
    rocprofv3 --pmc $SQSOLVE2 -d $OUT/sq_solve_${size}_${mode}b -o sq --output-format csv -- python3 $ROOT/scripts/run_solve_once.py $size $mode 4 > $OUT/sq_solve_${size}_${mode}b.log 2>&1
  done
done
if [ -x $ROOT/scripts/ubench_issue ] && [ "$TAG" = r02 ]; then
echo "[9] instruction issue costs"
$ROOT/scripts/ubench_issue > $OUT/ubench_issue.txt 2>&1
$ROOT/scripts/ubench_issue2 > $OUT/ubench_issue2.txt 2>&1
$ROOT/scripts/ubench_issue3 > $OUT/ubench_issue3.txt 2>&1
$ROOT/scripts/ubench_issue4 > $OUT/ubench_issue4.txt 2>&1
fi
find $OUT -name "*.csv" | wc -l
