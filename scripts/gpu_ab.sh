#!/bin/bash
# A/B of planar-kernel builds (interleaved rounds in one process): bash scripts/gpu_ab.sh libA.so libB.so [rounds]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab; mkdir -p "$OUT"
cd $ROOT
A=${1:-scripts/librtus_r01.so}; B=${2:-ray-tracing-ultrasound_amd/librtus.so}
timeout -k 10 300 python3 scripts/ab_planar.py $A $B ${3:-7} > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
grep -v "amdgpu.ids" $OUT/ab.txt
