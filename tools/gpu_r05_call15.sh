#!/bin/bash
# final: exit with live objects (both destroy orders), then the whole GPU suite and smoke() -- every step under its own timeout
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05l
mkdir -p $OUT
cd $R
for m in atexit reverse; do timeout 60 python tools/gpu_exit_leak.py $m 2>&1 | grep -v amdgpu.ids; echo "  rc=$?"; done | tee $OUT/exit_leak.log
timeout 380 python -m pytest tests -q -m gpu -x > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $OUT/gpu_tests.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
