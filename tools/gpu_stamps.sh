#!/bin/bash
# in-kernel stamp profile of k_tile_encode (diagnostic builds build_variants/lib_stamps*.so)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stamps
for v in stamps "$@"; do
JPEGAMD_LIB=$PWD/build_variants/lib_$v.so timeout -k 10 200 python tools/stamp_profile_tile.py $STAMP_ARGS > gpurun_out/stamps/$v.txt 2>&1 || { tail -20 gpurun_out/stamps/$v.txt; exit 1; }
echo "== $v"; tail -16 gpurun_out/stamps/$v.txt
done
