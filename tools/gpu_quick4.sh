#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-quick}
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_bench_objects.py -q -m gpu --maxfail=5 > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -v "^W2\|^E2" $OUT/tests.log | tail -40
