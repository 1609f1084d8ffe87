#!/bin/bash
# k_solve<3> diagnostics: per-panel shader-clock stamps, burst length / priority / scalar batching variants (libraries built in the container:
# python -m dsp_slam_amd.build --variant NAME -DFLAG ...)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05i
mkdir -p $OUT
cd $R
for v in ${VARIANTS:-"" _batch _batchst "" _batch}; do
  [ "$v" = "-" ] && v=""
  echo "== libdspgn$v.so"
  DSPGN_LIB=$R/dsp_slam_amd/lib/libdspgn$v.so timeout 200 python tools/gpu_small_loop.py 250 200 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/solve_variants.log
done
