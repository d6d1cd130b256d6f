#!/bin/bash
# build_head_variant.sh NAME [REV]: the library as committed at REV (default HEAD) -> build_variants/lib_NAME.so (the A/B baseline for uncommitted work)
set -e
NAME=$1; REV=${2:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C $ROOT archive $REV include jpeg-image-compression_amd | tar -x -C $T
make -s -C $T/jpeg-image-compression_amd > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
mkdir -p $ROOT/build_variants
cp $T/jpeg-image-compression_amd/libjpegamd.so $ROOT/build_variants/lib_$NAME.so
rm -rf $T
echo built build_variants/lib_$NAME.so from $REV
