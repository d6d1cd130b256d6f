#!/bin/bash
# build_ref_variant.sh NAME GIT_REV: library of an earlier commit as build_variants/lib_NAME.so (for same-box A/B runs)
set -e
NAME=$1; REV=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C $ROOT archive $REV jpeg-image-compression_amd include | tar -x -C $T
make -s -C $T/jpeg-image-compression_amd >/dev/null
mkdir -p $ROOT/build_variants
cp $T/jpeg-image-compression_amd/libjpegamd.so $ROOT/build_variants/lib_$NAME.so
rm -rf $T
echo built build_variants/lib_$NAME.so from $REV
