#!/bin/bash
# A/B of the planar kernel: round-1 build vs the working tree (interleaved rounds in one process)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab; mkdir -p "$OUT"
cd $ROOT
timeout -k 10 300 python3 scripts/ab_planar.py scripts/librtus_r01.so ray-tracing-ultrasound_amd/librtus.so ${1:-7} > $OUT/ab.txt 2>&1 || { tail -20 $OUT/ab.txt; exit 1; }
grep -v "amdgpu.ids" $OUT/ab.txt
