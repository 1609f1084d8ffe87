#!/bin/bash
# GPU box: the round's new tests first (all failures shown), then the whole GPU suite as the driver runs it.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04e}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_bench_objects.py tests/test_gpu_map_tools.py --maxfail=12 -q -m gpu > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"
tail -40 $OUT/new_tests.log
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -12 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
