#!/bin/bash
# GPU box, end of round: the whole GPU suite as the driver runs it + smoke, then the profile sequence.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$R/gpurun_out/${TAG}_tests
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/run_profiles.sh $TAG
