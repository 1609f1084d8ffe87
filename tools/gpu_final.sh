#!/bin/bash
# The round's closing GPU sequence (one gpurun call): smoke(), the whole -m gpu suite, the profile sequence.  Outputs under gpurun_out/<tag>/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rm -f gpurun_out/forensics_*.md
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -4 $OUT/gpu_tests.log
bash tools/run_profiles.sh $TAG > $OUT/run_profiles.log 2>&1; tail -12 $OUT/run_profiles.log | cut -c1-400
