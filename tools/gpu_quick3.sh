#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-quick}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu -k "direct or cluster or mixed" --maxfail=10 > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -v "^W2\|^E2" $OUT/tests.log | tail -15
