#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stamps
JPEGAMD_LIB=$PWD/build_variants/lib_stamps.so timeout -k 10 200 python tools/stamp_profile_tile.py > gpurun_out/stamps/stamps.txt 2>&1 || { tail -20 gpurun_out/stamps/stamps.txt; exit 1; }
tail -14 gpurun_out/stamps/stamps.txt
