#!/bin/bash
# A/B of library variants inside one gpurun call: tools/gpu_ab.sh <variant name> ...  (built by `python -m dsp_slam_amd.build --variant NAME FLAGS`)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab
mkdir -p $OUT
cd $R
for v in main "$@"; do
  if [ $v = main ]; then unset DSPGN_LIB; else export DSPGN_LIB=$R/dsp_slam_amd/lib/libdspgn_$v.so; fi
  timeout 300 python -m pytest tests/test_gpu_prepass.py -m gpu -q -k "every_mode or 64_cfg2 or without_the_audit or guard_fires" > $OUT/tests_$v.log 2>&1; tail -2 $OUT/tests_$v.log
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --latency-runs 3 --no-prepass-off 2> $OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  python - <<PY
import json
d=json.load(open("$OUT/bench_$v.json"))
print("$v", d["value"], "obj/s  K0 frac", d["prepass"]["frac"], "K0 ms/launch", d["prepass"]["avg_launch_ms"], " K1 frac", d["roofline"]["frac"], "lat", d.get("latency_ms_p50"), d.get("latency_kitti_size_ms_p50"), d["roofline"]["ms_per_step_by_kernel"])
PY
done
