#!/bin/bash
# librtus.so of a committed revision (default HEAD) -> variants/librtus_prev.so, for interleaved A/B runs against the working tree
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
T=$(mktemp -d)
git -C $ROOT archive $REV ray-tracing-ultrasound_amd/csrc include | tar -x -C $T
make -C $T/ray-tracing-ultrasound_amd/csrc -j4 > /dev/null
mkdir -p $ROOT/variants && cp $T/ray-tracing-ultrasound_amd/librtus.so $ROOT/variants/librtus_prev.so
rm -rf $T
echo $ROOT/variants/librtus_prev.so
