#!/bin/bash
# Experiment builds of librtus.so: scripts/build_variant.sh <name> "<extra hipcc flags>" -> variants/librtus_<name>.so
# (loaded through RTUS_LIB=...; *.so files are git-ignored but travel with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
EXTRA="$*"
SRC=$ROOT/ray-tracing-ultrasound_amd/csrc
OUT=$ROOT/variants; mkdir -p $OUT/obj_$NAME
COMMON="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -fno-fast-math -fno-slp-vectorize"
for f in rtus_shoot rtus_solve rtus_match rtus_fermat rtus_lens_fermat rtus_tfm rtus_capi; do
  fl=""
  [ $f = rtus_shoot ] && fl="-ffp-contract=off -mllvm -disable-machine-licm"
  [ $f = rtus_solve ] && fl="-ffp-contract=off -mllvm -disable-machine-licm"
  /opt/rocm/bin/hipcc $COMMON $fl $EXTRA -c $SRC/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj_$NAME/*.o -o $OUT/librtus_$NAME.so -Wl,-rpath,/opt/rocm/lib
rm -rf $OUT/obj_$NAME
echo $OUT/librtus_$NAME.so
