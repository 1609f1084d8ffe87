#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-quick}
mkdir -p $OUT
cd $R
timeout 300 python tools/gpu_small_loop.py 250 200 50 > $OUT/small_loop.txt 2>&1; grep -v "^W\|^E" $OUT/small_loop.txt | tail -5
