#!/bin/bash
# GPU box: the low-precision compute mode's tests, then the bench line (its lp_compute block) -- the quick check after a change to mlp_lpj_kernel.hip.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-lp}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_lp_compute.py tests/test_gpu_errors.py -q -m gpu -x > $OUT/lp_tests.log 2>&1; echo "lp tests rc=$?"; tail -3 $OUT/lp_tests.log
timeout 900 python bench.py --steps 20 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
r = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step", "value_fp32_only", "value_lp")})
lp = r["lp_compute"]
print(lp["ms_per_step"], lp["roofline"]["jacobian"], lp["roofline"]["ms_per_step_by_kernel"], lp["chained_result_vs_fp32_path"])
PY
