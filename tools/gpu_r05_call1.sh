#!/bin/bash
# Round 5, first GPU call: the new device-side Lie tests, the reworked cluster fallback + co-tenant tests, the 16-bit MFMA clock probe,
# one default bench line (per-leg rocprof_check, value_fp32_only).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_lie.py tests/test_gpu_round4.py tests/test_gpu_cotenant.py --maxfail=10 -q -m gpu -s > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"
tail -30 $OUT/new_tests.log
timeout 200 tools/probes/mfma16_probe.bin 3.0 > $OUT/mfma16_probe.md 2>&1; echo "probe rc=$?"; cat $OUT/mfma16_probe.md
timeout 600 python bench.py --steps 5 --warmup 1 2> $OUT/bench.err | tail -1 > $OUT/bench.json; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
