#!/bin/bash
# Round-4 evidence run, part 2: rocprofv3 kernel traces of the single-stream bench (8 images per launch and 1), SQ counter passes
# (one image per launch: comparable with round 2; pass 1 also at eight), HBM passes (FETCH_SIZE / WRITE_SIZE, separate runs) at one
# and at eight images per launch, in-kernel stamps of k_tile_encode (needs build_variants/lib_stamps.so).
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=r04 bash tools/gpu_r4_trace.sh || exit 1
TAG=r04_pmc1 IPL=1 PASSES="1 2 3 4 5" bash tools/gpu_r4_pmc.sh > gpurun_out/r04_pmc1.log 2>&1; tail -3 gpurun_out/r04_pmc1.log
TAG=r04_pmc8 IPL=8 PASSES="1 4 5" bash tools/gpu_r4_pmc.sh > gpurun_out/r04_pmc8.log 2>&1; tail -3 gpurun_out/r04_pmc8.log
[ -f build_variants/lib_stamps.so ] && bash tools/gpu_stamps.sh
