#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  ->  build_variants/lib_NAME.so  (objects in build_variants/obj_NAME; the shipped build is untouched)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/jpeg-image-compression_amd
OBJ=$ROOT/build_variants/obj_$NAME
rm -rf $OBJ; mkdir -p $OBJ
HIPFLAGS="--offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -O3 -std=c++17 -fPIC -I$ROOT/include -I$PKG/csrc $*"
for f in $PKG/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""
  [ $b = jpegamd_tile_pipeline ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
  /opt/rocm/bin/hipcc $HIPFLAGS $extra -c $f -o $OBJ/$b.o &
  pids="$pids $!"
  if [ $b = jpegamd_tile_pipeline ]; then
    /opt/rocm/bin/hipcc $HIPFLAGS $extra -DJPEGAMD_STAMPED_TU -c $f -o $OBJ/${b}_stamped.o &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p || { echo "build_variant $NAME: compile failed"; exit 1; }; done
others=$(ls $PKG/csrc/*.o $PKG/host/*.o | grep -v -E "csrc/jpegamd_(entropy|finalize|stitch|tile_pipeline|tile_pipeline_stamped)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/lib_$NAME.so $OBJ/*.o $others
echo built build_variants/lib_$NAME.so
