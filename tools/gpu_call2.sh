#!/bin/bash
# Round 4, GPU call 2: cluster kernel + new solve + gram; latency A/B + rocprof; bench; whole GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04b}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_bench_objects.py --maxfail=12 -q -m gpu > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"
tail -30 $OUT/new_tests.log
timeout 600 python tools/gpu_latency_ab.py 15 > $OUT/latency_ab.txt 2>&1; cat $OUT/latency_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat -o stats -- python $R/tools/gpu_small_loop.py 250 200 50 > $OUT/latency_run.txt 2>&1
DB=$(find /tmp/prof_lat -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/latency_kernel_stats.md 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat2 -o stats -- python $R/tools/gpu_small_loop.py 2000 500 20 > $OUT/latency_cfg2_run.txt 2>&1
DB=$(find /tmp/prof_lat2 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/latency_cfg2_kernel_stats.md 2>&1
cd $R
grep -v "^W2026\|^E2026" $OUT/latency_run.txt | tail -4; head -22 $OUT/latency_kernel_stats.md | cut -c1-170
timeout 600 python bench.py --steps 5 --warmup 1 2> $OUT/bench.err | tail -1 > $OUT/bench.json; python - <<PY
import json
try:
    b=json.load(open("$OUT/bench.json"))
    print({k:b.get(k) for k in ("value","ms_per_step","latency_ms_p50","latency_kitti_size_ms_p50","latency_one_shot_ms_p50")}, b["roofline"]["frac"], b["roofline"]["jac_kernel_frac"], b["prepass"]["frac"], b["roofline"]["ms_per_step_by_kernel"])
except Exception as e: print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -12 $OUT/gpu_tests.log
