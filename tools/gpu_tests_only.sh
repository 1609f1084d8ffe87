#!/bin/bash
# GPU box: the whole GPU suite as the driver runs it (without -x: every failure shown), then smoke().
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06}
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -25 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
