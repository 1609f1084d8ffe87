#!/bin/bash
# GPU box, short call: the round's exactness tests, the single-detection loop with k_solve's stage stamps, the latency A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-quick}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_configs.py -q -m gpu --maxfail=10 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 300 python tools/gpu_small_loop.py 250 200 50 > $OUT/small_loop.txt 2>&1; grep -v "^W\|^E" $OUT/small_loop.txt | tail -5
timeout 600 python tools/gpu_latency_ab.py 15 > $OUT/latency_ab.txt 2>&1; grep -v "^W\|^E" $OUT/latency_ab.txt | tail -26
