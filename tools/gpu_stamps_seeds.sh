#!/bin/bash
# stamp profile of k_tile_transform for the bench's three inputs (seeds 1000..1002)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stamps
for seed in 1000 1001 1002; do
JPEGAMD_LIB=$PWD/build_variants/lib_stamps.so timeout -k 10 200 python tools/stamp_profile_tile.py 8192 $seed > gpurun_out/stamps/seed$seed.txt 2>&1 || { tail -20 gpurun_out/stamps/seed$seed.txt; exit 1; }
echo "== seed $seed"; grep -E "exact fallback  |span|idle|finishing|per-wave total" gpurun_out/stamps/seed$seed.txt
done
