#!/bin/bash
# Round 5, fifth GPU call: the whole GPU suite as the driver runs it, smoke(), then the profile sequence (tools/run_profiles.sh r05 20 5).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05e
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/run_profiles.sh r05 20 5 2>&1 | tail -45
