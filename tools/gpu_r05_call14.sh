#!/bin/bash
# cluster exchange stores: plain (default where the placement census allows) against written through; the placement check's injection test
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05k
mkdir -p $OUT
cd $R
DSP_CLUSTER_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cotenant.py -m gpu -q -s -k "cluster or cotenant" > $OUT/tests.log 2>&1; grep -A12 "dspgn\]" $OUT/tests.log | head -60; tail -5 $OUT/tests.log
for i in 1 2; do timeout 120 python tools/gpu_detection_p50.py 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/p50.log; done
