#!/bin/bash
# K0 A/B: 128-point tiles as four waves x two column blocks (default) against eight waves x one column block (DSP_LP_WIDE=1: two waves per SIMD)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05g
mkdir -p $OUT
cd $R
for w in 0 1 0 1; do
  export DSP_LP_WIDE=$w
  echo "== DSP_LP_WIDE=$w"; timeout 200 python tools/probes/gpu_prepass_probe.py 2>&1 | grep "^f16" | tee -a $OUT/probe_wide$w.log
done
for w in 0 1; do
  export DSP_LP_WIDE=$w
  timeout 300 python -m pytest tests/test_gpu_prepass.py -m gpu -q -k "every_mode or 64_cfg2 or without_the_audit or guard_fires or small" > $OUT/tests_wide$w.log 2>&1; tail -2 $OUT/tests_wide$w.log
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --latency-runs 3 --no-prepass-off 2> $OUT/bench_wide$w.err | tail -1 > $OUT/bench_wide$w.json
  python - <<PY
import json
d=json.load(open("$OUT/bench_wide$w.json"))
print("wide=$w", d["value"], "obj/s  K0 frac", d["prepass"]["frac"], "K0 ms/launch", d["prepass"]["avg_launch_ms"], " K1 frac", d["roofline"]["frac"], "clock", d["prepass"].get("sustained_clock_mhz"), "lat", d.get("latency_ms_p50"), d.get("latency_kitti_size_ms_p50"), d["roofline"]["ms_per_step_by_kernel"])
PY
done
