#!/bin/bash
# build_full_variant.sh NAME -DFLAG...: the whole library (host files too) with extra defines, as build_variants/lib_NAME.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/jpeg-image-compression_amd
cp -r $ROOT/include $T/
(cd $ROOT/jpeg-image-compression_amd && tar -c --exclude='*.o' --exclude='*.so' --exclude=python . ) | tar -x -C $T/jpeg-image-compression_amd
make -s -C $T/jpeg-image-compression_amd EXTRA_HIPFLAGS="$*" CXX="g++ $*" > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
mkdir -p $ROOT/build_variants
cp $T/jpeg-image-compression_amd/libjpegamd.so $ROOT/build_variants/lib_$NAME.so
rm -rf $T
echo built build_variants/lib_$NAME.so
