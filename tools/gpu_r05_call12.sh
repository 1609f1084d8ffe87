#!/bin/bash
# k_solve: panel schedule (solver 3) + flattened load chain + wave-wide pose tail -- tests that pin it, the latency A/B, the stage stamps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05h
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_lie.py tests/test_gpu_forensics.py tests/test_gpu_bench_objects.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
timeout 300 python tools/gpu_latency_ab.py 15 > $OUT/latency_ab.log 2>&1; grep -v "^$" $OUT/latency_ab.log | head -40
timeout 200 python tools/gpu_small_loop.py 250 200 50 > $OUT/small_loop.log 2>&1; tail -8 $OUT/small_loop.log
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --latency-runs 15 --no-prepass-off 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], "obj/s  lat", d.get("latency_ms_p50"), d.get("latency_kitti_size_ms_p50"), d.get("latency_kitti_size_one_shot_ms_p50"), d["roofline"]["ms_per_step_by_kernel"])
PY
